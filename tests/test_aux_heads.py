"""Config-off heads that share the ConvDecoder kernels (LIDAR_SEG with top-k class-weighted cross entropy,
SEMANTIC_IMAGE, DEPTH — mile.py:337-363, trainer.py:132-182,338-365, losses.py:9-50) against the golden training step of
the REAL reference with those heads enabled (tests/golden/aux_b1s2.*, oracle/refimport/make_golden_aux.py).  CPU: the
oracle restatement.  GPU: the HIP model — 30 losses within 1e-3 relative, outputs and label pyramids within 2e-3, gradient
norms of the new decoders within 5e-3."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _fixture():
    return json.load(open(os.path.join(GOLD, 'aux_b1s2.json'))), np.load(os.path.join(GOLD, 'aux_b1s2_samples.npz'))


def _check_outputs(fx, smp, out, batch, tol):
    for k, st in fx['outputs'].items():
        t = batch[k[6:]] if k.startswith('batch.') else out[k]
        assert list(t.shape) == st['shape'], k
        f = t.detach().float().contiguous().view(-1)
        ref = torch.from_numpy(smp[('' if k.startswith('batch.') else 'out.') + k])
        got = f[::st['stride']][:ref.numel()].cpu()
        err = (got - ref).abs().max().item()
        assert err <= tol * max(st['absmean'], ref.abs().max().item(), 1e-6), f'{k}: {err}'


def test_oracle_aux_heads_match_reference():
    from muvo_amd.data.synthetic import make_aux_inputs, make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    model = R.MileRef(aux_heads=('lidar_seg', 'sem_image', 'depth'))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == fx['state_dict']
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'])
    batch.update(make_aux_inputs(b, s, fx['seed']))
    total, losses, out, pb = R.training_step(model, batch, eps, use_prior)
    assert set(losses) == set(fx['losses']) and len(losses) == 30
    for k, v in fx['losses'].items():
        assert abs(float(losses[k].detach()) - v) <= 2e-5 * max(abs(v), 1e-12), k
    _check_outputs(fx, smp, out, pb, 2e-4)


def test_oracle_segmentation_loss_properties():
    from oracle import muvo_ref as R
    torch.manual_seed(0)
    p, t = torch.randn(1, 2, 9, 6, 8), torch.randint(0, 9, (1, 2, 1, 6, 8))
    full = R._segmentation_loss(p, t, False, 0.5, False)
    assert torch.allclose(full, torch.nn.functional.cross_entropy(p.flatten(0, 1), t.flatten(0, 1)[:, 0]))
    # the mean over the hardest half is at least the mean over all pixels; ratio 1.0 reproduces it
    assert R._segmentation_loss(p, t, True, 0.5, False) >= full
    assert torch.allclose(R._segmentation_loss(p, t, True, 1.0, False), full)


@pytest.mark.gpu
def test_hip_aux_heads_match_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_aux_inputs, make_batch, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _fixture()
    b, s = fx['b'], fx['s']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
    cfg.LIDAR_SEG.ENABLED = cfg.SEMANTIC_IMAGE.ENABLED = cfg.DEPTH.ENABLED = True
    for name in ('LIDAR_SEG', 'SEMANTIC_IMAGE'):
        for key in ('N_CLASSES', 'USE_TOP_K', 'TOP_K_RATIO', 'USE_WEIGHTS'):
            assert getattr(getattr(cfg, name), key) == fx['cfg'][name][key], (name, key)
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.train()
    tr.preprocess.augment = False
    assert {k: list(v.shape) for k, v in tr.model.state_dict().items()} == fx['state_dict']
    detinit.fill_state_dict_(tr.model)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    opts, _ = tr.configure_optimizers()          # the new decoders live in the flat parameter store too
    eps, use_prior = make_noise(b, s, seed=fx['seed'])
    batch = make_batch(b, s, seed=fx['seed'], device=dev)
    batch.update(make_aux_inputs(b, s, fx['seed'], device=dev))
    opts[0].zero_grad()
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
    total = tr.loss_reducing(losses)
    total.backward()
    assert set(losses) == set(fx['losses']) and len(losses) == 30
    for k, v in fx['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-3 * max(abs(v), 1e-12), (k, losses[k].item(), v)
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
def test_hip_segmentation_loss_kernel(dev):
    """muvo_seg_ce_{fwd,bwd} against F.cross_entropy (weights, top-k, a class that never occurs) at a full range-view size."""
    import torch.nn.functional as F
    from muvo_amd.losses import VOXEL_SEG_WEIGHTS, SegmentationLoss
    torch.manual_seed(0)
    p = torch.randn(2, 3, 9, 64, 256)
    t = torch.randint(0, 8, (2, 3, 1, 64, 256))            # class 8 absent
    for top_k, weights in ((True, True), (False, True), (False, False)):
        pc = p.clone().requires_grad_(True)
        w = torch.tensor(VOXEL_SEG_WEIGHTS) if weights else None
        ref = F.cross_entropy(pc.flatten(0, 1), t.flatten(0, 1)[:, 0], reduction='none', weight=w).view(2, 3, -1)
        if top_k:
            ref = ref.topk(int(0.5 * ref.shape[2]), dim=-1)[0]
        ref = ref.mean()
        ref.backward()
        pg = p.to(dev).requires_grad_(True)
        got = SegmentationLoss(use_top_k=top_k, top_k_ratio=0.5, use_weights=weights, is_bev=False)(pg, t.to(dev))
        got.backward()
        assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
        assert float((pg.grad.cpu() - pc.grad).abs().max()) <= 1e-5 * float(pc.grad.abs().max())
