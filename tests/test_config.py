"""CPU: config surface parity with the reference's muvo/config.py (defaults dumped from the reference:
tests/golden/default_cfg.json; effective base_1d: tests/golden/effective_cfg_base_1d.json)."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _norm(x):
    if isinstance(x, dict):
        return {k: _norm(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_norm(v) for v in x]
    return x


def test_defaults_equal_reference_defaults():
    from muvo_amd.config import get_cfg
    ref = json.load(open(os.path.join(GOLD, 'default_cfg.json')))
    assert _norm(get_cfg().convert_to_dict()) == _norm(ref)


def test_base_1d_effective_config():
    from muvo_amd.config import base_1d_cfg
    ref = json.load(open(os.path.join(GOLD, 'effective_cfg_base_1d.json')))
    got = _norm(base_1d_cfg(RECEPTIVE_FIELD=ref['RECEPTIVE_FIELD'], FUTURE_HORIZON=ref['FUTURE_HORIZON']).convert_to_dict())
    for k in ('TAG', 'DATASET'):
        got.pop(k), ref.pop(k)
    assert got == _norm(ref)


def test_parser_file_and_overrides():
    from muvo_amd import config
    path = os.path.join(os.path.dirname(config.__file__), 'configs', 'test_base_1d.yml')
    args = config.get_parser().parse_args(['--config-file', path, 'BATCHSIZE', '2', 'MODEL.TRANSFORMER.CHANNELS', '256'])
    cfg = config.get_cfg(args)
    assert cfg.BATCHSIZE == 2 and cfg.MODEL.TRANSFORMER.CHANNELS == 256 and cfg.RECEPTIVE_FIELD == 6
    assert cfg.is_frozen()
    with pytest.raises(AttributeError):
        cfg.BATCHSIZE = 3


def test_unknown_keys():
    from muvo_amd import config
    cfg = config.get_cfg(cfg_dict={'NOT_A_KEY': 1, 'MODEL': {'ALSO_NEW': 2}})  # dict: tolerated with a warning
    assert cfg.NOT_A_KEY == 1 and cfg.MODEL.ALSO_NEW == 2
    with pytest.raises(KeyError):
        config.get_cfg().merge_from_list(['NOPE', '1'])
    c = config.get_cfg()
    c._merge({'CML_DATASET_VERSION': '2', 'LOSSES': {'PERCEPTUAL': {'ENABLED': False}}})  # 2-D-branch keys: dropped
    assert 'CML_DATASET_VERSION' not in c
    with pytest.raises(KeyError):
        c._merge({'LOSSES': {'NOPE': 1}})


def test_scope_guard():
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.models.mile import Mile
    with pytest.raises(NotImplementedError):
        Mile(base_1d_cfg(**{'MODEL.TRANSFORMER.LARGE': True}))          # configurations outside the built rows fail loudly
    with pytest.raises(NotImplementedError):
        Mile(base_1d_cfg(**{'MODEL.LIDAR.POINT_PILLAR.ENABLED': True}))
