"""Gradient fixture of the REAL reference at the BASELINE workload itself (runs ONLY in the build container).

base_1d, batch 2 x seq_len 10, full sizes: WorldModelTrainer.forward -> compute_loss -> backward of the imported reference
(muvo/trainer.py:213-231,251-402) in training mode (train-mode BatchNorm over the 20 frames, dropout off, augmentation off,
explicit RSSM noise), float32.  The activations of 20 frames do not fit the container's memory, so the decoders, the
encoders and the voxel-decoder blocks are recomputed in backward (torch.utils.checkpoint around the REAL modules' forward:
deterministic CPU arithmetic, the gradients are the same numbers; BatchNorm's running statistics move twice, which no
gradient depends on).  Written: the 21 losses (they must equal the forward-only fixture base1d_b2s10_fwd.json), the L2 norm
of all 440 parameter gradients, 1024 strided samples of the ten largest gradient tensors and of six named ones.

Writes tests/golden/base1d_b2s10_bwd.json, base1d_b2s10_bwd_samples.npz.
Usage: python oracle/refimport/make_golden_bwd.py --fp64   (2 min for the float32 pass, 14 min and 48.5 GB for the float64 one; the
committed fixture was written with --fp64: keys grad_l2_fp64 / grad_l2_ref32_err / grad64.*)"""
import argparse
import json
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402

NAMED = ['type_embedding', 'rssm.recurrent_model.weight_hh', 'voxel_decoder.constant_tensor', 'encoder.conv1.weight',
         'transformer_encoder.layers.0.self_attn.in_proj_weight', 'voxel_decoder.conv3.conv2.conv_act.0.weight',
         'range_view_encoder.conv1.weight', 'rgb_decoder.trans_conv3.0.weight', 'encoder.layer1.0.bn1.weight']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=2)
    ap.add_argument('--s', type=int, default=10)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--fp64', action='store_true',
                    help='also run the same step of the reference in float64 (convolutions fed one frame at a time: the im2col '
                         'buffer of torch\'s float64 CPU convolution is per frame) and record, per tensor, the L2 norm of the float64 '
                         'gradient and the distance of the float32 gradient from it (the reference\'s own rounding noise at this size)')
    args = ap.parse_args()
    b, s, seed = args.b, args.s, args.seed
    tag = f'b{b}s{s}_bwd'
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    trainer = ref_trainer.WorldModelTrainer(cfg)
    trainer.train()
    trainer.preprocess.eval()
    model = trainer.model
    detinit.fill_state_dict_(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    from torch.utils.checkpoint import checkpoint
    vd = model.voxel_decoder
    mods = [model.rgb_decoder, model.lidar_re, model.encoder, model.range_view_encoder, model.feat_decoder,
            model.range_view_decoder, vd.first_conv, *vd.middle_conv, vd.conv1, vd.conv2, vd.conv3]
    for mod in mods:
        mod.forward = (lambda *a, _f=mod.forward: checkpoint(_f, *a, use_reentrant=False))
    eps, use_prior = make_noise(b, s, seed=seed)
    coin = detinit.uniform_01(detinit.name_key(f'noise:{seed}') + 7, s)
    batch = make_batch(b, s, seed=seed)
    t0 = time.time()
    with G.NoisePatch(eps, coin):
        output, _ = trainer.forward(batch)
    losses = trainer.compute_loss(batch, output)
    total = trainer.loss_reducing(losses)
    print(f'reference forward + losses {time.time() - t0:.1f}s total={float(total):.7f}', flush=True)
    t0 = time.time()
    total.backward()
    print(f'reference backward {time.time() - t0:.1f}s, peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GB',
          flush=True)
    rec = dict(total=float(total), losses={k: float(v) for k, v in losses.items()})
    fwd = os.path.join(G.REPO, 'tests', 'golden', f'base1d_b{b}s{s}_fwd.json')
    if os.path.exists(fwd):
        g = json.load(open(fwd))['steps'][0]
        dev = max(abs(rec['losses'][k] - v) / max(abs(v), 1e-12) for k, v in g['losses'].items())
        print(f'losses vs the forward-only fixture: max rel deviation {dev:.2e}')
        rec['losses_vs_fwd_fixture'] = dev
        assert dev < 1e-6
    params = dict(model.named_parameters())
    gn = {n: (None if p.grad is None else float(p.grad.double().pow(2).sum().sqrt())) for n, p in params.items()}
    rec['grad_l2'] = gn
    assert sum(v is not None for v in gn.values()) == 440, sum(v is not None for v in gn.values())
    rec['grad_absmax'] = {n: (None if p.grad is None else float(p.grad.abs().max())) for n, p in params.items()}
    samples = {}
    big = sorted([n for n in gn if gn[n] is not None], key=lambda n: -params[n].numel())
    for n in big[:10] + [n for n in NAMED if n in params and n not in big[:10]]:
        _, smp = G.tensor_stats(params[n].grad)
        samples['grad.' + n] = smp
    if args.fp64:
        g32 = {n: p.grad.detach().clone() for n, p in params.items() if p.grad is not None}
        del output, losses, total, trainer, model, params
        import gc
        gc.collect()
        t0 = time.time()
        torch.set_default_dtype(torch.float64)
        _tensor_float = torch.Tensor.float
        torch.Tensor.float = lambda self, *a, **k: self.double()   # the reference calls .float() on inputs
        try:
            tr64 = ref_trainer.WorldModelTrainer(cfg)
            tr64.train()
            tr64.preprocess.eval()
            detinit.fill_state_dict_(tr64.model)          # the same closed-form initial weights (BatchNorm buffers at their defaults)
            tr64.double()
            for m in tr64.model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
                if isinstance(m, torch.nn.MultiheadAttention):
                    m.dropout = 0.0

            def per_frame(fwd):
                return lambda x: torch.cat([fwd(x[i:i + 1]) for i in range(x.shape[0])], 0) if x.shape[0] > 1 else fwd(x)
            for m in tr64.model.modules():
                if isinstance(m, (torch.nn.Conv3d, torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    m.forward = per_frame(m.forward)
            vd = tr64.model.voxel_decoder
            mods = [tr64.model.rgb_decoder, tr64.model.lidar_re, tr64.model.encoder, tr64.model.range_view_encoder,
                    tr64.model.feat_decoder, tr64.model.range_view_decoder, vd.first_conv, *vd.middle_conv, vd.conv1, vd.conv2, vd.conv3]
            for mod in mods:
                mod.forward = (lambda *a, _f=mod.forward: checkpoint(_f, *a, use_reentrant=False))
            b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in make_batch(b, s, seed=seed).items()}
            with G.NoisePatch(eps.double(), coin):
                out64, _ = tr64.forward(b64)
            l64 = tr64.compute_loss(b64, out64)
            tot64 = tr64.loss_reducing(l64)
            tot64.backward()
        finally:
            torch.set_default_dtype(torch.float32)
            torch.Tensor.float = _tensor_float
        p64 = dict(tr64.model.named_parameters())
        rec['total_fp64'] = float(tot64)
        rec['grad_l2_fp64'] = {n: (None if p64[n].grad is None else float(p64[n].grad.pow(2).sum().sqrt())) for n in gn}
        rec['grad_l2_ref32_err'] = {n: float((g32[n].double() - p64[n].grad).pow(2).sum().sqrt()) for n in g32}
        for key in list(samples):
            n = key[5:]
            g64 = p64[n].grad.detach().contiguous().view(-1)
            stride = max(1, g64.numel() // 1024)
            samples['grad64.' + n] = g64[::stride][:1024].clone().numpy()
        print(f'float64 reference fwd+bwd {time.time() - t0:.1f}s total={float(tot64):.9f} (float32 {rec["total"]:.9f}), '
              f'peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GB', flush=True)
    fixture = dict(tag=tag, b=b, s=s, seed=seed, use_prior=use_prior, steps=[rec],
                   note='float32 backward of the imported reference, decoders / encoders / voxel-decoder blocks checkpointed')
    gold = os.path.join(G.REPO, 'tests', 'golden')
    with open(os.path.join(gold, f'base1d_{tag}.json'), 'w') as f:
        json.dump(fixture, f)
    np.savez_compressed(os.path.join(gold, f'base1d_{tag}_samples.npz'), **samples)
    print('wrote fixtures for', tag)


if __name__ == '__main__':
    main()
