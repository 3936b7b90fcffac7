"""Config surface of the reference (muvo/config.py:1-369): `get_parser()`, `get_cfg(args, cfg_dict)`,
`CfgNode.convert_to_dict()`, every default key — on an own minimal yacs-style node (fvcore/yacs are not
dependencies).  Differences, on purpose: YAML files may carry the three keys of the reference's unreleased 2-D
branch (TOLERATED_EXTRA_KEYS); they are accepted and dropped instead of raising KeyError."""
import argparse
import copy
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
TOLERATED_EXTRA_KEYS = ('CML_DATASET_VERSION', 'MODEL.TRANSFORMER_TRANSITION', 'LOSSES.PERCEPTUAL')
# Extension keys (NOT in the reference, absent from the defaults so that the default config stays equal to the reference's;
# accepted from YAML files, cfg dicts and command-line overrides, and kept):
#   MODEL.CONSTANT_SIZE.{RGB,LIDAR,VOXEL} - the seed sizes the reference hard-codes as (5, 13), (1, 16) and (3, 3, 1)
#   (muvo/models/mile.py:322-336,391-396): every decoder output is 64x its seed, so (1, 32) gives the 64 x 2048 range view and
#   (4, 4, 1) the 256 x 256 x 64 voxel grid BASELINE.json names.  Parity for non-default values: own oracle only ("unpinned").
EXTENSION_KEYS = ('MODEL.CONSTANT_SIZE', 'MODEL.CONSTANT_SIZE.RGB', 'MODEL.CONSTANT_SIZE.LIDAR', 'MODEL.CONSTANT_SIZE.VOXEL')
DEFAULT_CONSTANT_SIZE = {'RGB': (5, 13), 'LIDAR': (1, 16), 'VOXEL': (3, 3, 1)}


def constant_sizes(cfg):
    """(rgb, lidar, voxel) seed sizes of the three decoders: MODEL.CONSTANT_SIZE.* if given, else the reference's constants."""
    cs = cfg.MODEL.get('CONSTANT_SIZE', None) or {}
    out = []
    for k in ('RGB', 'LIDAR', 'VOXEL'):
        v = tuple(int(x) for x in cs.get(k, DEFAULT_CONSTANT_SIZE[k]))
        if len(v) != len(DEFAULT_CONSTANT_SIZE[k]) or min(v) < 1:
            raise ValueError(f'MODEL.CONSTANT_SIZE.{k} must be {len(DEFAULT_CONSTANT_SIZE[k])} positive integers, got {v}')
        out.append(v)
    return tuple(out)


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, '_frozen', False)
        object.__setattr__(self, '_new_allowed', False)
        for k, v in (init or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) else v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError(f'Attempted to set {name} on a frozen CfgNode')
        self[name] = value

    def _children(self):
        return [v for v in self.values() if isinstance(v, CfgNode)]

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        object.__setattr__(self, '_frozen', True)
        for c in self._children():
            c.freeze()

    def defrost(self):
        object.__setattr__(self, '_frozen', False)
        for c in self._children():
            c.defrost()

    def is_frozen(self):
        return self._frozen

    def set_new_allowed(self, flag):
        object.__setattr__(self, '_new_allowed', flag)
        for c in self._children():
            c.set_new_allowed(flag)

    def convert_to_dict(self):
        return {k: (v.convert_to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def _merge(self, other, path=''):
        for k, v in other.items():
            full = f'{path}{k}'
            if k not in self:
                if full in TOLERATED_EXTRA_KEYS:
                    continue
                if self._new_allowed or full in EXTENSION_KEYS:
                    dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) else copy.deepcopy(v))
                    continue
                raise KeyError(f'Non-existent config key: {full}')
            cur = self[k]
            if isinstance(cur, CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f'config key {full} is a section, got {type(v).__name__}')
                cur._merge(v, full + '.')
            else:
                if isinstance(cur, tuple) and isinstance(v, list):
                    v = tuple(v)
                elif isinstance(cur, list) and isinstance(v, tuple):
                    v = list(v)
                elif isinstance(cur, float) and isinstance(v, int) and not isinstance(v, bool):
                    v = float(v)
                dict.__setitem__(self, k, copy.deepcopy(v))

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_file(self, path):
        with open(path) as f:
            data = yaml.safe_load(f) or {}
        base = data.pop('_BASE_', None)
        if base:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(os.path.abspath(path)), base)
            self.merge_from_file(base)
        self._merge(data)

    def merge_from_list(self, opts):
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError('override list must be KEY VALUE pairs')
        for key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for i, p in enumerate(parts[:-1]):
                if p not in node and '.'.join(parts[:i + 1]) in EXTENSION_KEYS:
                    dict.__setitem__(node, p, CfgNode())
                node = node[p]
            if parts[-1] not in node and key not in EXTENSION_KEYS:
                raise KeyError(f'Non-existent config key: {key}')
            val = raw
            if isinstance(raw, str):
                try:
                    val = yaml.safe_load(raw)
                except yaml.YAMLError:
                    val = raw
            node._merge({parts[-1]: val}, '.'.join(parts[:-1]) + ('.' if len(parts) > 1 else ''))


CN = CfgNode


def _load_defaults():
    with open(os.path.join(_HERE, 'configs', 'defaults.yml')) as f:
        d = yaml.safe_load(f)
    node = CfgNode(d)
    # tuple-typed defaults of the reference
    node.IMAGE['SIZE'] = tuple(node.IMAGE.SIZE)
    node.IMAGE['IMAGENET_MEAN'] = tuple(node.IMAGE.IMAGENET_MEAN)
    node.IMAGE['IMAGENET_STD'] = tuple(node.IMAGE.IMAGENET_STD)
    for k in ('AUGMENTATION_TRANSLATE', 'AUGMENTATION_SCALE', 'AUGMENTATION_SHEAR'):
        node.ROUTE[k] = tuple(node.ROUTE[k])
    return node


_C = _load_defaults()


def get_parser():
    parser = argparse.ArgumentParser(description='World model training')
    parser.add_argument('--config-file', default='', metavar='FILE', help='path to config file')
    parser.add_argument('opts', help='Modify config options using the command-line', default=None,
                        nargs=argparse.REMAINDER)
    return parser


def _extra_keys(known, other, path=''):
    out = []
    for k, v in other.items():
        full = f'{path}.{k}' if path else k
        if k not in known:
            if full not in EXTENSION_KEYS:
                out.append(full)
        elif isinstance(known[k], dict) and isinstance(v, dict):
            out.extend(_extra_keys(known[k], v, full))
    return sorted(out)


def get_cfg(args=None, cfg_dict=None):
    """Defaults, then cfg_dict (unknown keys tolerated with a warning), then args.config_file / args.opts (frozen)."""
    cfg = _C.clone()
    if cfg_dict is not None:
        extra = _extra_keys(cfg, cfg_dict)
        if extra:
            print(f'Warning - the cfg_dict merging into the main cfg has keys that do not exist in main: {extra}')
            cfg.set_new_allowed(True)
        cfg.merge_from_other_cfg(cfg_dict)
    if args is not None:
        if args.config_file:
            cfg.merge_from_file(args.config_file)
        cfg.merge_from_list(args.opts)
        cfg.freeze()
    return cfg


def base_1d_cfg(**overrides):
    """Effective base_1d configuration (defaults <- configs/muvo.yml <- configs/test_base_1d.yml) + overrides."""
    cfg = _C.clone()
    cfg.merge_from_file(os.path.join(_HERE, 'configs', 'test_base_1d.yml'))
    flat = []
    for k, v in overrides.items():
        flat += [k.replace('__', '.'), v]
    cfg.merge_from_list(flat)
    return cfg
