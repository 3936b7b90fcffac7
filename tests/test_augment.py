"""Training-time input augmentation of PreProcess.forward (preprocess.py:45-48,213-214,295-367) against
tests/golden/augment.*, which oracle/refimport/make_golden_aug.py produced by running the REAL PreProcess in training mode.

CPU: (i) the host-side draw (muvo_amd/augment.py) repeats the reference's RNG call sequence — with the fixture's seed it must
reproduce the fixture's parameter tables bit for bit; (ii) the oracle restatement fed with those tables reproduces the
reference's augmented tensors.  -m gpu: the HIP kernels (csrc/augment.hip) through the product PreProcess with the tables
as explicit inputs: route maps bit-exact (nearest resampling: index work), images within 2e-6 absolute (the blur's 25-term
sums and the contrast mean are float reductions); the label pyramid is untouched; frames whose draw says "nothing" are
bit-identical to the augmentation-free pipeline."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _fixture():
    return json.load(open(os.path.join(GOLD, 'augment.json'))), np.load(os.path.join(GOLD, 'augment.npz'))


def _cfg(fx):
    from muvo_amd.config import base_1d_cfg
    cfg = base_1d_cfg()
    for grp, kv in fx['overrides'].items():
        for k, v in kv.items():
            cfg[grp][k] = v
    return cfg


def _check(fx, smp, image, label1, route, tol):
    stride = int(smp['sample_stride'])
    for name, t, key in (('image', image, 'image_sample'), ('rgb_label_1', label1, 'label_sample')):
        got = t.detach().float().cpu()
        ref = torch.from_numpy(smp[key])
        err = (got.flatten()[::stride] - ref).abs().max().item()
        scale = 1.0 if name == 'rgb_label_1' else 1.0 / 0.224      # the normalised image is the label divided by std
        assert err <= tol * scale, f'{name}: sample max err {err:.3e}'
        d = got.double().flatten(2)
        for f, (s_ref, l_ref) in enumerate(zip(fx[name]['sum'], fx[name]['l2'])):
            n = d.shape[-1]
            assert abs(d.sum(-1).flatten()[f].item() - s_ref) <= tol * scale * n, (name, f)
            assert abs(d.pow(2).sum(-1).sqrt().flatten()[f].item() - l_ref) <= 1e-5 * max(l_ref, 1.0), (name, f)
    return (route.detach().float().cpu() != torch.from_numpy(smp['route_map'])).sum().item()


def test_host_draws_reproduce_reference_rng_sequence():
    from muvo_amd import augment
    fx, smp = _fixture()
    cfg = _cfg(fx)
    torch.manual_seed(fx['seed'])
    pix = augment.draw_pixel_params(cfg, fx['b'], fx['s'])
    route = augment.draw_route_params(cfg, fx['b'], cfg.ROUTE.SIZE)
    assert np.array_equal(pix.numpy(), smp['pixel_params']) and np.array_equal(route.numpy(), smp['route_params'])
    # default probabilities: roughly 60 % of the frames are touched, a route map rarely
    torch.manual_seed(0)
    pix = augment.draw_pixel_params(_cfg({'overrides': {}}), 50, 10)
    assert 0.45 < float(((pix[:, 0] != 0) | (pix[:, 2] != 0)).float().mean()) < 0.85
    assert ((pix[:, 7:10][pix[:, 2] == 1] >= 0.7) & (pix[:, 7:10][pix[:, 2] == 1] <= 1.3)).all()
    assert (pix[:, 10].abs() <= 0.1).all()


def test_oracle_augmentation_matches_reference():
    from muvo_amd.data.synthetic import make_aug_batch
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    raw = make_aug_batch(fx['b'], fx['s'], fx['seed'])
    out = R.preprocess(raw, R.base_1d_cfg(), pixel_aug=torch.from_numpy(smp['pixel_params']),
                       route_aug=torch.from_numpy(smp['route_params']))
    assert _check(fx, smp, out['image'], out['rgb_label_1'], out['route_map'], 1e-6) == 0


@pytest.mark.gpu
def test_hip_augmentation_matches_reference(dev):
    from muvo_amd.data.synthetic import make_aug_batch
    from muvo_amd.models.preprocess import PreProcess
    fx, smp = _fixture()
    cfg = _cfg(fx)
    pre = PreProcess(cfg)
    pre.train()
    raw = make_aug_batch(fx['b'], fx['s'], fx['seed'], device=dev)
    batch = dict(raw)
    batch['_pixel_aug'] = torch.from_numpy(smp['pixel_params'])
    batch['_route_aug'] = torch.from_numpy(smp['route_params'])
    out = pre(batch)
    assert out['rgb_label_1'].data_ptr() != out['image'].data_ptr()
    mism = _check(fx, smp, out['image'], out['rgb_label_1'], out['route_map'], 2e-6)
    print(f'route-map pixels that differ from the reference: {mism} of {out["route_map"].numel()}')
    assert mism == 0
    # augmentation-free pipeline: identical label pyramid; untouched frames / samples are bit-identical
    pre.augment = False
    plain = pre(dict(raw))
    pix, route = smp['pixel_params'], smp['route_params']
    for k in ('rgb_label_2', 'rgb_label_4', 'range_view_label_1', 'voxel_label_2'):
        assert torch.equal(plain[k], out[k]), k
    b, s = fx['b'], fx['s']
    for f in range(b * s):
        same = torch.equal(plain['image'][f // s, f % s], out['image'][f // s, f % s])
        if pix[f, 0] == 0 and pix[f, 2] == 0:      # (the converse need not hold: a blur with sigma ~0.1 changes nothing)
            assert same, f
    for i in range(b):
        assert torch.equal(plain['route_map'][i], out['route_map'][i]) == (route[i, 0] == 0), i


@pytest.mark.gpu
def test_training_step_draws_its_own_augmentation(dev):
    """training mode without explicit tables: the draws come from torch's CPU generator (same seed -> same step), the
    augmented image is what the encoder sees, and `augment = False` / eval() switch it off."""
    from muvo_amd.data.synthetic import make_aug_batch
    from muvo_amd.models.preprocess import PreProcess
    fx, _ = _fixture()
    pre = PreProcess(_cfg(fx))
    pre.train()
    raw = make_aug_batch(2, 3, fx['seed'], device=dev)
    torch.manual_seed(7)
    a = pre(dict(raw))
    torch.manual_seed(7)
    b_ = pre(dict(raw))
    torch.manual_seed(8)
    c = pre(dict(raw))
    assert torch.equal(a['image'], b_['image']) and torch.equal(a['route_map'], b_['route_map'])
    assert not torch.equal(a['image'], c['image'])
    pre.eval()
    d = pre(dict(raw))
    assert not torch.equal(a['image'], d['image'])
    lo, hi = d['rgb_label_1'].min().item(), d['rgb_label_1'].max().item()
    assert 0.0 <= lo and hi <= 1.0 and a['rgb_label_1'].min().item() >= 0.0 and a['rgb_label_1'].max().item() <= 1.0
