"""Evaluation metrics (SSIM, PSNR, Chamfer distance, SSC IoU — muvo/metrics.py, trainer.py:426-490).
CPU: the oracle restatement against the golden statistics of the REAL reference classes (tests/golden/metrics.json,
oracle/refimport/make_golden_metrics.py).  GPU: the HIP kernels through muvo_amd.metrics against the same fixture and the
oracle; integer counts bit-exact, floating statistics within 1e-4 relative (fp32 reference arithmetic; the HIP kernels
accumulate in fp64)."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'metrics.json')


def _check(st, g, rel):
    for k in ('ssim', 'psnr', 'cd'):
        assert abs(float(st[k]) - g[k]) <= rel * abs(g[k]), (k, float(st[k]), g[k])
    for k in ('precision', 'recall', 'iou', 'iou_ssc_mean'):
        assert abs(float(st[k]) - g['ssc'][k]) <= 1e-6 * max(abs(g['ssc'][k]), 1e-12), (k, st[k], g['ssc'][k])
    for k in ('completion', 'tps', 'fps', 'fns'):
        assert [int(v) for v in st[k]] == g['ssc'][k], (k, st[k], g['ssc'][k])
    assert torch.allclose(torch.as_tensor(st['iou_ssc']).float().cpu(), torch.tensor(g['ssc']['iou_ssc']), rtol=1e-6)


def test_oracle_metrics_match_reference():
    from muvo_amd.data.metric_inputs import metric_case
    from oracle import muvo_ref
    fx = json.load(open(GOLD))
    m = muvo_ref.EvalMetrics(fx['n_classes'], fx['scale'])
    for k, g in enumerate(fx['after_batch']):
        m.add_batch(**metric_case(k))
        _check(m.stats(), g, 2e-6)


def test_oracle_metric_edge_cases():
    from oracle import muvo_ref
    # all voxels ignored / nothing predicted: the reference reports zeros instead of dividing by zero (metrics.py:102-110)
    m = muvo_ref.EvalMetrics(2)
    comp, tps, fps, fns = muvo_ref.ssc_counts(torch.zeros(1, 4, 4, 2, dtype=torch.long),
                                              torch.full((1, 4, 4, 2), 255, dtype=torch.uint8), 2)
    assert comp.tolist() == [0, 0, 0] and tps.tolist() == [0, 0]
    assert m.stats()['iou'] == 0.0
    # identical images: SSIM = 1, PSNR = inf
    x = torch.rand(1, 1, 3, 16, 20)
    assert abs(float(muvo_ref.ssim_frames(x, x).mean()) - 1.0) < 1e-5
    assert torch.isinf(muvo_ref.psnr_frames(x, x)).all()
    # Chamfer distance of a point set with itself is 0; symmetric in its arguments
    p, q = torch.rand(2, 17, 3), torch.rand(2, 9, 3)
    assert float(muvo_ref.chamfer_frames(p, p).abs().max()) < 1e-3
    assert torch.allclose(muvo_ref.chamfer_frames(p, q), muvo_ref.chamfer_frames(q, p))


@pytest.mark.gpu
def test_hip_metrics_match_reference(dev):
    from muvo_amd.data.metric_inputs import metric_case
    from muvo_amd.metrics import EvalMetrics
    from oracle import muvo_ref
    fx = json.load(open(GOLD))
    m, o = EvalMetrics(fx['n_classes'], fx['scale']), muvo_ref.EvalMetrics(fx['n_classes'], fx['scale'])
    for k, g in enumerate(fx['after_batch']):
        c = metric_case(k)
        o.add_batch(**c)
        m.add_batch(**{kk: (v.to(dev) if torch.is_tensor(v) else torch.from_numpy(v).to(dev)) for kk, v in c.items()})
        _check(m.stats(), g, 1e-4)
        so, sm = o.stats(), m.stats()
        for kk in ('ssim', 'psnr', 'cd'):
            assert abs(float(sm[kk]) - float(so[kk])) <= 1e-4 * abs(float(so[kk]))


@pytest.mark.gpu
def test_hip_metrics_full_size_properties(dev):
    """base_1d sizes (20 frames of 320x832 rgb, 10000 Chamfer points, 192x192x64 voxels): properties that do not need
    the oracle at full size."""
    from muvo_amd import metrics as M
    torch.manual_seed(0)
    x = torch.rand(2, 10, 3, 320, 832, device=dev)
    assert torch.allclose(M.ssim_frames(x, x), torch.ones(20, device=dev), atol=1e-5)
    y = (x + 0.1).clamp(0, 1)
    s_xy, s_yx = M.ssim_frames(x, y), M.ssim_frames(y, x)
    assert torch.allclose(s_xy, s_yx, rtol=1e-5) and float(s_xy.max()) < 1.0          # SSIM is symmetric
    ps = M.psnr_frames(x, y)
    mse = ((x - y) ** 2).mean(dim=(2, 3, 4))
    assert torch.allclose(ps, 20 * torch.log10(1.0 / mse.sqrt()), rtol=1e-5)
    p, q = torch.rand(20, 10000, 3, device=dev) * 50, torch.rand(20, 10000, 3, device=dev) * 50
    assert float(M.chamfer_frames(p, p).abs().max()) == 0.0
    assert torch.allclose(M.chamfer_frames(p, q), M.chamfer_frames(q, p), rtol=1e-6)
    sub = M.chamfer_frames(p[:2, :512], q[:2, :512])
    d = torch.cdist(p[:2, :512].double(), q[:2, :512].double())
    ref = (d.min(1)[0].mean(1) + d.min(2)[0].mean(1)) / 2
    assert torch.allclose(sub.double(), ref, rtol=1e-5)
    logits = torch.randn(20, 2, 192, 192, 64, device=dev)
    label = (torch.rand(20, 192, 192, 64, device=dev) < 0.1).to(torch.uint8)
    label[:, :3] = 255
    comp, tps, fps, fns = M.ssc_counts(logits, label, 2)
    valid = int((label != 255).sum())
    assert int(tps.sum() + fps.sum()) == valid and int(tps.sum() + fns.sum()) == valid    # every valid voxel is counted once
    assert int(comp[0]) == int(tps[1]) and int(comp[1]) == int(fps[1]) and int(comp[2]) == int(fns[1])   # 2 classes
    pred = logits.argmax(1)
    assert int(tps[1]) == int(((pred == 1) & (label == 1)).sum())
