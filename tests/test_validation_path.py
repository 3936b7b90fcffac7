"""Validation / imagination path (trainer.py:232-249 shared_step(mode='val'),
mile.py:771-850 Mile.imagine, transition.py:151-173 imagine_step) against the fixture generated from the REAL
reference (oracle/refimport/make_golden_val.py): reconstruction of the first RECEPTIVE_FIELD frames, then
PREDICTION.N_SAMPLES imagined roll-outs of FUTURE_HORIZON steps with the recorded actions, scored with the same 20
losses.  CPU: the oracle restatement; GPU: the HIP path (through the C ABI)."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TAG = 'b1r2f2'


def _fixture():
    fx = json.load(open(os.path.join(GOLD, f'base1d_val_{TAG}.json')))
    smp = np.load(os.path.join(GOLD, f'base1d_val_{TAG}_samples.npz'))
    return fx, smp


def _check_losses(fx, losses, losses_im, tol):
    assert set(losses) == set(fx['losses']) and len(losses) == 21
    for k, v in fx['losses'].items():
        assert abs(float(losses[k]) - v) <= tol * max(abs(v), 1e-6), (k, float(losses[k]), v)
    assert len(losses_im) == fx['n_samples']
    for li, ref in zip(losses_im, fx['losses_imagine']):
        assert set(li) == set(ref) and len(li) == 20 and 'probabilistic' not in li   # no prior/posterior in imagined outputs
        for k, v in ref.items():
            assert abs(float(li[k]) - v) <= tol * max(abs(v), 1e-6), (k, float(li[k]), v)


def _check_outputs(fx, smp, out, outs_im, rtol):
    for key, st in fx['outputs'].items():
        tag, name = key.split('.', 1)
        t = out[name] if tag == 'rf' else outs_im[int(tag[2:])][name]
        assert list(t.shape) == st['shape'], key
        f = t.detach().float().contiguous().view(-1).cpu()
        ref = torch.from_numpy(smp[key])
        got = f[::st['stride']][:ref.numel()]
        err = (got - ref).abs().max().item()
        assert err <= rtol * max(st['absmean'], ref.abs().max().item(), 1e-6), f'{key}: {err}'


def test_oracle_validation_path_matches_reference():
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    b, rf, fh, ns = fx['b'], fx['rf'], fx['fh'], fx['n_samples']
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    model = R.MileRef()
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(b, rf + ns * fh, seed=fx['seed'])
    assert use_prior[:rf] == fx['use_prior']
    batch = make_batch(b, rf + fh, seed=fx['seed'])
    losses, out, losses_im, outs_im = R.validation_step(model, batch, rf, fh, eps, use_prior, ns)
    _check_losses(fx, losses, losses_im, 2e-5)
    _check_outputs(fx, smp, out, outs_im, 2e-4)


@pytest.mark.gpu
def test_hip_validation_path_matches_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _fixture()
    b, rf, fh, ns = fx['b'], fx['rf'], fx['fh'], fx['n_samples']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=rf, FUTURE_HORIZON=fh, STEPS=100000)
    assert cfg.PREDICTION.N_SAMPLES == ns
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.preprocess.augment = False   # validation_step calls self.train() like the reference (trainer.py:405), which would
    detinit.fill_state_dict_(tr.model)  # switch the input augmentation on; the fixture was made without it
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    eps, use_prior = make_noise(b, rf + ns * fh, seed=fx['seed'])
    batch = make_batch(b, rf + fh, seed=fx['seed'], device=dev)
    res, losses, out, losses_im, outs_im = tr.validation_step(batch, noise=eps.to(dev), use_prior=use_prior)
    assert set(res) == {'val0_loss', 'val0_loss_imagine'}
    _check_losses(fx, {k: v.item() for k, v in losses.items()}, [{k: v.item() for k, v in li.items()} for li in losses_im],
                  1e-3)
    _check_outputs(fx, smp, out, outs_im, 2e-3)
    ref_total = sum(fx['losses'].values())
    assert abs(res['val0_loss'].item() - ref_total) <= 1e-3 * ref_total
    ref_im = sum(sum(li.values()) for li in fx['losses_imagine']) / ns
    assert abs(res['val0_loss_imagine'].item() - ref_im) <= 1e-3 * ref_im


@pytest.mark.gpu
def test_hip_validation_metrics_match_oracle(dev):
    """validation_step also feeds the evaluation metrics (trainer.py:415-417): reconstruction metrics on the observed frames,
    imagination metrics on the future frames.  The HIP metric kernels are compared with the oracle restatement of
    muvo/metrics.py evaluated on the very tensors the HIP model produced (integer counts bit-exact, statistics 1e-4)."""
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, _ = _fixture()
    b, rf, fh, ns = fx['b'], fx['rf'], fx['fh'], fx['n_samples']
    cfg = base_1d_cfg(RECEPTIVE_FIELD=rf, FUTURE_HORIZON=fh, STEPS=100000)
    tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
    tr.preprocess.augment = False   # validation_step calls self.train() like the reference (trainer.py:405), which would
    detinit.fill_state_dict_(tr.model)  # switch the input augmentation on; the fixture was made without it
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    eps, use_prior = make_noise(b, rf + ns * fh, seed=fx['seed'])
    batch = make_batch(b, rf + fh, seed=fx['seed'], device=dev)
    cd_index = (detinit.hash_u64(detinit.name_key('cd_index'), 10000) % np.uint64(64 * 1024)).astype(np.int64)
    _, _, out, _, outs_im = tr.validation_step(batch, noise=eps.to(dev), use_prior=use_prior, cd_index=cd_index)
    for metrics, sl, outputs in ((tr.metrics_vals[0], slice(0, rf), [out]), (tr.metrics_vals_imagine[0], slice(rf, rf + fh), outs_im)):
        assert set(metrics) == {'ssim', 'psnr', 'cd', 'ssc'}
        o = R.EvalMetrics(2, cfg.LIDAR_RE.SCALE)
        for od in outputs:
            o.add_batch(od['rgb_1'].float().cpu(), batch['rgb_label_1'][:, sl].float().cpu(),
                        od['lidar_reconstruction_1'].float().cpu(), batch['range_view_label_1'][:, sl].float().cpu(), cd_index,
                        od['voxel_1'].float().cpu(), batch['voxel_label_1'][:, sl, 0].cpu())
        so = o.stats()
        assert abs(float(metrics['ssim'].get_stat()) - so['ssim']) <= 1e-4 * abs(so['ssim'])
        assert abs(float(metrics['psnr'].get_stat()) - so['psnr']) <= 1e-4 * abs(so['psnr'])
        assert abs(float(metrics['cd'].get_stat()) - so['cd']) <= 1e-4 * abs(so['cd'])
        ssc = metrics['ssc']
        assert [ssc.completion_tp, ssc.completion_fp, ssc.completion_fn] == so['completion']
        assert ssc.tps.tolist() == so['tps'] and ssc.fps.tolist() == so['fps'] and ssc.fns.tolist() == so['fns']
        assert abs(ssc.get_stats()['iou'] - so['iou']) < 1e-9
