"""Loss options that are off in base_1d (leftovers): VoxelLoss with class weights and the top-k selection
(muvo/losses.py:144-186; needs the 9-class voxel head), LOSSES.RGB_INSTANCE (instance-masked second RGB term, muvo/trainer.py:
303-321, preprocess.py:115-125) and RegressionLoss(norm=2) (muvo/losses.py:53-71) - against the golden training step of the REAL
reference with those options on (tests/golden/lossopts_b1s2.*, oracle/refimport/make_golden_lossopts.py).  CPU: the oracle
restatement.  GPU: the HIP model - 21 losses within 1e-3 relative, outputs and mask pyramids within 2e-3, gradient norms of the
voxel and RGB decoders within 5e-3."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _fixture():
    return json.load(open(os.path.join(GOLD, 'lossopts_b1s2.json'))), np.load(os.path.join(GOLD, 'lossopts_b1s2_samples.npz'))


def _check_outputs(fx, smp, out, batch, tol):
    for k, st in fx['outputs'].items():
        t = batch[k[6:]] if k.startswith('batch.') else out[k]
        assert list(t.shape) == st['shape'], k
        f = t.detach().float().contiguous().view(-1)
        ref = torch.from_numpy(smp[('' if k.startswith('batch.') else 'out.') + k])
        got = f[::st['stride']][:ref.numel()].cpu()
        err = (got - ref).abs().max().item()
        assert err <= tol * max(st['absmean'], ref.abs().max().item(), 1e-6), f'{k}: {err}'


def test_oracle_loss_options_match_reference():
    from muvo_amd.data.synthetic import make_batch, make_image_instance_mask, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    v = fx['cfg']['VOXEL_SEG']
    cfg = dict(R.base_1d_cfg(), VOXEL_N_CLASSES=v['N_CLASSES'], VOXEL_USE_WEIGHTS=v['USE_WEIGHTS'], VOXEL_USE_TOP_K=v['USE_TOP_K'],
               VOXEL_TOP_K_RATIO=v['TOP_K_RATIO'], RGB_INSTANCE=True)
    model = R.MileRef(cfg)
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'], n_voxel_classes=9)
    batch['image_instance_mask'] = make_image_instance_mask(b, s, fx['seed'])
    with torch.no_grad():
        total, losses, out, pb = R.training_step(model, batch, eps, use_prior)
    assert set(losses) == set(fx['losses']) and len(losses) == 21
    for k, val in fx['losses'].items():
        assert abs(float(losses[k]) - val) <= 2e-5 * max(abs(val), 1e-12), k
    _check_outputs(fx, smp, out, pb, 2e-4)


@pytest.mark.gpu
def test_hip_loss_options_match_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch, make_image_instance_mask, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    v = fx['cfg']['VOXEL_SEG']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
    cfg.VOXEL_SEG.N_CLASSES, cfg.VOXEL_SEG.USE_WEIGHTS = v['N_CLASSES'], v['USE_WEIGHTS']
    cfg.VOXEL_SEG.USE_TOP_K, cfg.VOXEL_SEG.TOP_K_RATIO = v['USE_TOP_K'], v['TOP_K_RATIO']
    cfg.LOSSES.RGB_INSTANCE = True
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.train()
    tr.preprocess.augment = False
    detinit.fill_state_dict_(tr.model)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    opts, _ = tr.configure_optimizers()
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'], n_voxel_classes=9, device=dev)
    batch['image_instance_mask'] = make_image_instance_mask(b, s, fx['seed'], device=dev)
    opts[0].zero_grad()
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
    total = tr.loss_reducing(losses)
    total.backward()
    assert set(losses) == set(fx['losses']) and len(losses) == 21
    for k, val in fx['losses'].items():
        assert abs(losses[k].item() - val) <= 1e-3 * max(abs(val), 1e-12), (k, losses[k].item(), val)
    assert abs(total.item() - fx['total']) <= 1e-3 * fx['total']
    _check_outputs(fx, smp, output, batch, 2e-3)
    params = dict(tr.model.named_parameters())
    bad = []
    for n, ref in fx['grad_l2'].items():
        got = params[n].grad.double().pow(2).sum().sqrt().item()
        if abs(got - ref) > 5e-3 * max(ref, 1e-12) + 1e-7:
            bad.append(f'{n} x{got / max(ref, 1e-30):.4f}')
    assert not bad, '; '.join(bad)


@pytest.mark.gpu
def test_hip_loss_classes(dev):
    """the reference's loss-module API on the kernels: VoxelLoss (weights / top-k), SpatialRegressionLoss with an instance mask
    (incl. the empty mask -> 0), RegressionLoss(norm=2) against plain PyTorch, values and gradients."""
    import torch.nn.functional as F
    from muvo_amd.losses import VOXEL_SEG_WEIGHTS, RegressionLoss, SpatialRegressionLoss, VoxelLoss
    torch.manual_seed(1)
    p = torch.randn(1, 2, 9, 24, 24, 8)
    t = torch.randint(0, 8, (1, 2, 1, 24, 24, 8))
    for top_k, weights in ((True, True), (False, True), (True, False)):
        pc = p.clone().requires_grad_(True)
        w = torch.tensor(VOXEL_SEG_WEIGHTS) if weights else None
        ref = F.cross_entropy(pc.flatten(0, 1), t.flatten(0, 1)[:, 0], reduction='none', weight=w).view(1, 2, -1)
        if top_k:
            ref = ref.topk(int(0.25 * ref.shape[2]), dim=-1)[0]
        ref = ref.mean()
        ref.backward()
        pg = p.to(dev).requires_grad_(True)
        got = VoxelLoss(use_top_k=top_k, top_k_ratio=0.25, use_weights=weights)(pg, t.to(torch.uint8).to(dev))
        got.backward()
        assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()), (top_k, weights)
        assert float((pg.grad.cpu() - pc.grad).abs().max()) <= 1e-5 * float(pc.grad.abs().max())
    pred, tgt = torch.randn(2, 3, 3, 40, 52), torch.rand(2, 3, 3, 40, 52)
    mask = torch.rand(2, 3, 1, 40, 52) < 0.2
    for m in (mask, torch.zeros_like(mask)):
        pc = pred.clone().requires_grad_(True)
        ref = (pc - tgt).abs().sum(2, keepdim=True)[m].mean() if m.any() else pc.sum() * 0
        ref.backward()
        pg = pred.to(dev).requires_grad_(True)
        got = SpatialRegressionLoss(norm=1)(pg, tgt.to(dev), instance_mask=m.to(dev))
        got.backward()
        assert abs(got.item() - ref.item()) <= 1e-5 * max(abs(ref.item()), 1e-6)
        assert float((pg.grad.cpu() - pc.grad).abs().max()) <= 1e-6 + 1e-5 * float(pc.grad.abs().max())
    a, bt = torch.randn(2, 5, 3), torch.randn(2, 5, 3)
    ac = a.clone().requires_grad_(True)
    ref = ((ac - bt) ** 2).sum(-1, keepdim=True).mean()
    ref.backward()
    ag = a.to(dev).requires_grad_(True)
    got = RegressionLoss(norm=2)(ag, bt.to(dev))
    got.backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * ref.item()
    assert float((ag.grad.cpu() - ac.grad).abs().max()) <= 1e-5 * float(ac.grad.abs().max())
