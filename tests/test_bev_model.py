"""BEV-lifting model variant (MODEL.TRANSFORMER.BEV=True — Decoder with bilinear upsampling, mono depth
head, FrustumPooling, bev_down_sample_4; mile.py:33-59,506-524) against the golden training step of the REAL reference
(tests/golden/bev_b1s2.*, oracle/refimport/make_golden_bev.py).  CPU: the oracle restatement.  GPU: the HIP model —
losses within 1e-3 relative, outputs within 2e-3, gradient norms of the BEV-specific parameters within 5e-3 + 2x the reference's OWN
gradient-norm change under a 4e-6 relative perturbation of its conv outputs (fixture field rounding_sensitivity: up to
1.5e-2 for this b1s2 step, because ReLU / L1 sign decisions flip), on the exact-fp32 kernels and on the default bf16x3
policy."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _fixture():
    return json.load(open(os.path.join(GOLD, 'bev_b1s2.json'))), np.load(os.path.join(GOLD, 'bev_b1s2_samples.npz'))


def _check_outputs(fx, smp, out, tol):
    for k, st in fx['outputs'].items():
        t = out['posterior']['mu'] if k == 'posterior.mu' else out[k]
        assert list(t.shape) == st['shape'], k
        f = t.detach().float().contiguous().view(-1)
        ref = torch.from_numpy(smp['out.' + k])
        got = f[::st['stride']][:ref.numel()].cpu()
        err = (got - ref).abs().max().item()
        assert err < tol * max(st['absmean'], ref.abs().max().item(), 1e-6), f'{k}: {err}'


def test_oracle_bev_step_matches_reference():
    from muvo_amd.data.frustum_inputs import camera_pose
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    model = R.MileRef(bev=True)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == fx['state_dict']
    bev_intr = model.frustum_pooling.bev_intrinsics.clone()
    detinit.fill_state_dict_(model)
    model.frustum_pooling.bev_intrinsics.copy_(bev_intr)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'])
    batch['extrinsics'] = camera_pose(b, s)
    total, losses, out, _ = R.training_step(model, batch, eps, use_prior)
    for k, v in fx['losses'].items():
        assert abs(float(losses[k]) - v) <= 2e-5 * max(abs(v), 1e-12), k
    _check_outputs(fx, smp, out, 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('arith', ['f32', 'policy'])
def test_hip_bev_step_matches_reference(dev, arith):
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_F32 if arith == 'f32' else ops.CONV_BF16X3, min_gflop=-1.0)
    try:
        _hip_bev_step(dev, arith)
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


def _hip_bev_step(dev, arith):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.frustum_inputs import camera_pose
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
    cfg.MODEL.TRANSFORMER.BEV = True
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.train()
    tr.preprocess.augment = False
    assert {k: list(v.shape) for k, v in tr.model.state_dict().items()} == fx['state_dict']
    bev_intr = tr.model.frustum_pooling.bev_intrinsics.clone()
    detinit.fill_state_dict_(tr.model)
    tr.model.frustum_pooling.bev_intrinsics.copy_(bev_intr)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'], device=dev)
    batch['extrinsics'] = camera_pose(b, s).to(dev)
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
    total = tr.loss_reducing(losses)
    total.backward()
    assert set(losses) == set(fx['losses'])
    for k, v in fx['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-3 * max(abs(v), 1e-12), (k, losses[k].item(), v)
    assert abs(total.item() - fx['total']) <= 1e-3 * fx['total']
    _check_outputs(fx, smp, output, 2e-3)
    params = dict(tr.model.named_parameters())
    bad = []
    # Both arithmetics: this b1s2 step sits next to a decision whose flip rescales every BEV-branch gradient by ~1 % (the
    # exact-fp32 run lands on either side from run to run through float-atomic summation order alone; the reference shows
    # the same jump under a 4e-6 perturbation), so the bar is 5e-3 + 2x the reference's measured sensitivity.
    rtol = 5e-3 + 2.0 * max(fx['rounding_sensitivity']['grad_l2'].values())
    for n, ref in fx['grad_l2'].items():
        got = params[n].grad.double().pow(2).sum().sqrt().item()
        if abs(got - ref) > rtol * max(ref, 1e-12) + 1e-7:
            bad.append((n, got, ref))
    assert not bad, '; '.join(f'{n} x{g / r:.4f}' for n, g, r in bad)
