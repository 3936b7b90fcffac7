"""SSIM as a training loss (LOSSES.SSIM=True — trainer.py:312-318, SSIMLoss losses.py:292-348) against the
golden training step of the REAL reference (tests/golden/ssim_b1s2.json, oracle/refimport/make_golden_ssim.py): the oracle
on CPU, the HIP model on GPU (24 losses within 1e-3, RGB-decoder gradient norms within 5e-3), and the kernel pair
muvo_ssim_maps / muvo_ssim_bwd against autograd through the oracle's formula."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ssim_b1s2.json')


def test_oracle_ssim_loss_step_matches_reference():
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx = json.load(open(GOLD))
    b, s = fx['b'], fx['s']
    model = R.MileRef(cfg={**R.base_1d_cfg(), 'SSIM': True})
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    total, losses, _, _ = R.training_step(model, make_batch(b, s, seed=fx['seed']), eps, use_prior)
    assert set(losses) == set(fx['losses']) and len(losses) == 24
    for k, v in fx['losses'].items():
        assert abs(float(losses[k].detach()) - v) <= 2e-5 * max(abs(v), 1e-12), k


@pytest.mark.gpu
def test_hip_ssim_loss_kernels(dev):
    from muvo_amd.losses import SSIMLoss
    from oracle import muvo_ref as R
    torch.manual_seed(0)
    for shape in ((1, 2, 3, 40, 57), (2, 1, 3, 80, 208), (1, 1, 1, 11, 11)):
        t = torch.rand(*shape)
        p = (t + 0.2 * torch.randn(*shape)).clamp(0, 1)
        pc = p.clone().requires_grad_(True)
        ref = R.ssim_frames(pc, t).mean()
        ref.backward()
        pg = p.to(dev).requires_grad_(True)
        got = SSIMLoss(channel=shape[2])(pg, t.to(dev))
        got.backward()
        assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()), shape
        assert float((pg.grad.cpu() - pc.grad).abs().max()) <= 2e-4 * float(pc.grad.abs().max()), shape


@pytest.mark.gpu
def test_hip_ssim_loss_step_matches_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx = json.load(open(GOLD))
    b, s = fx['b'], fx['s']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
    cfg.LOSSES.SSIM = True
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.train()
    tr.preprocess.augment = False
    detinit.fill_state_dict_(tr.model)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'], device=dev)
    losses, _, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
    total = tr.loss_reducing(losses)
    total.backward()
    assert set(losses) == set(fx['losses']) and len(losses) == 24
    for k, v in fx['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-3 * max(abs(v), 1e-12), (k, losses[k].item(), v)
    params = dict(tr.model.named_parameters())
    bad = [f'{n} x{params[n].grad.double().pow(2).sum().sqrt().item() / max(ref, 1e-30):.4f}' for n, ref in fx['grad_l2'].items()
           if abs(params[n].grad.double().pow(2).sum().sqrt().item() - ref) > 5e-3 * max(ref, 1e-12) + 1e-7]
    assert not bad, '; '.join(bad)
