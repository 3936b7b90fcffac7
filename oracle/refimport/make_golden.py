"""Golden-fixture generator (runs ONLY in the build container, where /root/reference exists).

Imports the real reference (`/root/reference/muvo`) through the import stubs in
oracle/refimport/stubs, loads deterministic weights (muvo_amd.utils.detinit), runs
WorldModelTrainer.forward -> compute_loss -> backward -> AdamW x2 on a seeded synthetic batch
with explicit RSSM noise, and writes small fixtures to tests/golden/.  It also checks the
oracle restatement (oracle/muvo_ref.py) against the reference on the same inputs and records
the observed deviations in the fixture.

Usage: python oracle/refimport/make_golden.py [--b 1 --s 2 --tag b1s2]
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
sys.path.insert(0, REPO)

from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, 'stubs'))
    import muvo.trainer as ref_trainer  # noqa
    import muvo.config as ref_config  # noqa
    return ref_trainer, ref_config


def effective_cfg_dict(ref_config, names=('muvo.yml', 'test_base_1d.yml')):
    """defaults <- muvo.yml <- test_base_1d.yml merged as dicts, unknown keys pruned."""
    base = ref_config._C.clone().convert_to_dict()

    def merge(dst, src, path=''):
        for k, v in src.items():
            if k == '_BASE_':
                continue
            if k not in dst:
                print(f'  pruned unknown key {path}{k}')
                continue
            if isinstance(dst[k], dict) and isinstance(v, dict):
                merge(dst[k], v, path + k + '.')
            else:
                dst[k] = tuple(v) if isinstance(dst[k], tuple) and isinstance(v, list) else v
    for n in names:
        with open(os.path.join(REF, 'muvo', 'configs', n)) as f:
            merge(base, yaml.safe_load(f))
    base['PRETRAINED']['PATH'] = ''
    return base


class NoisePatch:
    """Feeds torch.randn_like / torch.rand calls of the reference RSSM from explicit tensors."""

    def __init__(self, eps, coin_values):
        self.eps, self.coins = eps, list(coin_values)
        self.i = 0
        self.j = 0

    def __enter__(self):
        self._rl, self._r = torch.randn_like, torch.rand
        s = self.eps.shape[1]

        def randn_like(x, *a, **k):
            t, which = divmod(self.i, 2)
            self.i += 1
            assert t < s
            return self.eps[:, t, which].to(x.dtype)

        def rand(*a, **k):
            v = self.coins[self.j]
            self.j += 1
            return torch.tensor([v])
        torch.randn_like, torch.rand = randn_like, rand
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.rand = self._rl, self._r


def tensor_stats(t: torch.Tensor, nsample=1024):
    t = t.detach().float().contiguous().view(-1)
    n = t.numel()
    stride = max(1, n // nsample)
    d = t.double()
    return dict(shape=None, mean=d.mean().item(), absmean=d.abs().mean().item(), l2=d.pow(2).sum().sqrt().item(),
                min=d.min().item(), max=d.max().item(), stride=stride), t[::stride][:nsample].clone().numpy()


def argmax_digest(v: torch.Tensor):
    """voxel logits (b,s,C,X,Y,Z) -> sha256 of packed argmax bits + per-frame popcounts (bit-exact target)."""
    am = v.argmax(dim=2).to(torch.uint8).contiguous().numpy()
    h = hashlib.sha256(np.packbits(am.astype(bool)).tobytes() if am.max() <= 1 else am.tobytes()).hexdigest()
    pops = [int(x) for x in am.reshape(am.shape[0] * am.shape[1], -1).sum(1)]
    return h, pops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=1)
    ap.add_argument('--s', type=int, default=2)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--tag', default=None)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--skip-oracle', action='store_true')
    ap.add_argument('--fp64-checkpoint', action='store_true',
                    help='float64 run: recompute the three decoders in backward (b*s > 4 does not fit 62 GB otherwise)')
    ap.add_argument('--no-fp64', action='store_true', help='skip the float64 run of the reference (gradient noise floor)')
    ap.add_argument('--threads', type=int, default=8, help='CPU threads (another count = another summation order in the reference\'s own reductions)')
    args = ap.parse_args()
    tag = args.tag or f'b{args.b}s{args.s}'
    torch.manual_seed(0)
    torch.set_num_threads(args.threads)

    ref_trainer, ref_config = import_reference()
    cfg_dict = effective_cfg_dict(ref_config)
    cfg_dict['RECEPTIVE_FIELD'] = args.s
    cfg_dict['FUTURE_HORIZON'] = 0
    cfg_dict['STEPS'] = 100000
    t0 = time.time()
    trainer = ref_trainer.WorldModelTrainer(cfg_dict)
    trainer.train()
    trainer.preprocess.eval()  # augmentation off
    model = trainer.model
    detinit.fill_state_dict_(model)
    for m in model.modules():  # dropout off (SURVEY 8c iii)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    print(f'reference built in {time.time() - t0:.1f}s; params '
          f'{sum(p.numel() for p in model.parameters()) / 1e6:.3f} M')

    spec = {k: list(v.shape) for k, v in model.state_dict().items()}
    param_names = [n for n, _ in model.named_parameters()]
    os.makedirs(os.path.join(REPO, 'tests', 'golden'), exist_ok=True)
    if tag == 'b1s2':
        with open(os.path.join(REPO, 'tests', 'golden', 'state_dict_spec.json'), 'w') as f:
            json.dump({'state_dict': spec, 'parameters': param_names}, f)

    # keep a copy of the effective cfg (data, not code)
    if tag == 'b1s2':
        with open(os.path.join(REPO, 'tests', 'golden', 'effective_cfg_base_1d.json'), 'w') as f:
            json.dump(cfg_dict, f, indent=1, default=list)

    opts, scheds = trainer.configure_optimizers()
    opt, sched = opts[0], scheds[0]['scheduler']
    eps, use_prior = make_noise(args.b, args.s, seed=args.seed)
    coin = detinit.uniform_01(detinit.name_key(f'noise:{args.seed}') + 7, args.s)

    fixture = dict(tag=tag, b=args.b, s=args.s, seed=args.seed, use_prior=use_prior, steps=[])
    samples = {}
    oracle_model = None
    if not args.skip_oracle:
        from oracle import muvo_ref
        oracle_model = muvo_ref.MileRef()
        missing = oracle_model.load_state_dict(model.state_dict(), strict=True)
        print('oracle load_state_dict strict OK', missing)
        oracle_model.train()
        oracle_model.set_dropout(0.0)
        oracle_opt, oracle_sched = muvo_ref.make_optimizer(oracle_model, oracle_model.cfg)

    for step in range(args.steps):
        batch = make_batch(args.b, args.s, seed=args.seed + step)
        raw = {k: v.clone() for k, v in batch.items()}
        t0 = time.time()
        with NoisePatch(eps, coin):
            output, _ = trainer.forward(batch)
        losses = trainer.compute_loss(batch, output)
        total = trainer.loss_reducing(losses)
        opt.zero_grad(set_to_none=True)
        total.backward()
        print(f'step {step}: reference fwd+bwd {time.time() - t0:.1f}s total={total.item():.6f}')
        rec = dict(total=float(total.item()), losses={k: float(v.item()) for k, v in losses.items()})
        if step == 0:
            outs = {}
            for k in ['rgb_1', 'rgb_2', 'rgb_4', 'lidar_reconstruction_1', 'lidar_reconstruction_2',
                      'lidar_reconstruction_4', 'voxel_1', 'voxel_2', 'voxel_4', 'throttle_brake', 'steering']:
                st, smp = tensor_stats(output[k])
                st['shape'] = list(output[k].shape)
                outs[k] = st
                samples['out.' + k] = smp
            for grp in ('prior', 'posterior'):
                for k in ('hidden_state', 'sample', 'mu', 'sigma'):
                    st, smp = tensor_stats(output[grp][k])
                    st['shape'] = list(output[grp][k].shape)
                    outs[f'{grp}.{k}'] = st
                    samples[f'out.{grp}.{k}'] = smp
            for k in ['rgb_label_2', 'rgb_label_4', 'range_view_label_4', 'voxel_label_2', 'voxel_label_4', 'image',
                      'route_map']:
                st, smp = tensor_stats(batch[k])
                st['shape'] = list(batch[k].shape)
                outs['batch.' + k] = st
                samples['batch.' + k] = smp
            rec['outputs'] = outs
            h, pops = argmax_digest(output['voxel_1'])
            rec['voxel_1_argmax_sha256'] = h
            rec['voxel_1_argmax_popcounts'] = pops
            # argmax can only be bit-exact where the decision margin exceeds fp32 summation-order noise: record the
            # voxels whose top-2 logit margin is below NEAR_TIE and a digest of the argmax with those voxels zeroed
            NEAR_TIE = 2e-3
            v1 = output['voxel_1'].detach()
            top2 = v1.topk(2, dim=2).values
            margin = (top2[:, :, 0] - top2[:, :, 1]).reshape(-1)
            tie = torch.nonzero(margin < NEAR_TIE).reshape(-1).to(torch.int32)
            am = v1.argmax(dim=2).reshape(-1).to(torch.uint8).clone()
            am[tie.long()] = 0
            samples['voxel_1_near_tie_idx'] = tie.numpy()
            samples['voxel_1_near_tie_argmax'] = v1.argmax(dim=2).reshape(-1).to(torch.uint8)[tie.long()].numpy()
            rec['voxel_1_near_tie_margin'] = NEAR_TIE
            rec['voxel_1_near_tie_count'] = int(tie.numel())
            rec['voxel_1_argmax_sha256_excl_near_ties'] = hashlib.sha256(
                np.packbits(am.numpy().astype(bool)).tobytes()).hexdigest()
            rec['voxel_1_margin_min'] = float((output['voxel_1'][:, :, 0] - output['voxel_1'][:, :, 1]).abs().min())
            gn = {}
            for n, p in model.named_parameters():
                gn[n] = None if p.grad is None else float(p.grad.double().pow(2).sum().sqrt())
            rec['grad_l2'] = gn
            big = sorted([n for n in gn if gn[n] is not None], key=lambda n: -dict(model.named_parameters())[n].numel())
            for n in big[:10] + ['type_embedding', 'rssm.recurrent_model.weight_hh',
                                 'voxel_decoder.constant_tensor', 'encoder.conv1.weight',
                                 'transformer_encoder.layers.0.self_attn.in_proj_weight',
                                 'voxel_decoder.conv3.conv2.conv_act.0.weight']:
                _, smp = tensor_stats(dict(model.named_parameters())[n].grad)
                samples['grad.' + n] = smp
        if oracle_model is not None:
            from oracle import muvo_ref
            t0 = time.time()
            o_total, o_losses, o_out, _ = muvo_ref.training_step(oracle_model, raw, eps, use_prior)
            oracle_opt.zero_grad(set_to_none=True)
            o_total.backward()
            dev = {k: abs(float(o_losses[k]) - rec['losses'][k]) / max(abs(rec['losses'][k]), 1e-12) for k in rec['losses']}
            gdev = 0.0
            refp = dict(model.named_parameters())
            for n, p in oracle_model.named_parameters():
                g = refp[n].grad
                assert (g is None) == (p.grad is None), n
                if g is not None:
                    gdev = max(gdev, float((p.grad - g).norm() / (g.norm() + 1e-30)))
            odev = max(float((o_out[k] - output[k]).abs().max()) for k in ['rgb_1', 'lidar_reconstruction_1', 'voxel_1'])
            print(f'  oracle fwd+bwd {time.time() - t0:.1f}s: max rel loss dev {max(dev.values()):.3e}, '
                  f'max rel grad dev {gdev:.3e}, max abs out dev {odev:.3e}')
            rec['oracle_vs_reference'] = dict(max_rel_loss_dev=max(dev.values()), max_rel_grad_dev=gdev,
                                              max_abs_out_dev=odev)
            oracle_opt.step()
            oracle_sched.step()
            del o_total, o_losses, o_out
        del output, losses, total
        gc.collect()
        try:                      # hand the freed float32 activations back to the OS before the float64 run
            import ctypes
            ctypes.CDLL('libc.so.6').malloc_trim(0)
        except OSError:
            pass
        if step == 0 and not args.no_fp64:
            import resource
            print(f'  peak RSS before the fp64 run {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GB')
            # The same step of the REAL reference in float64: the truth the fp32 gradients are measured against.
            # tests compare |hip - ref64| with the reference's own fp32 rounding error |ref32 - ref64| per tensor.
            t0 = time.time()
            torch.set_default_dtype(torch.float64)
            _tensor_float = torch.Tensor.float
            torch.Tensor.float = lambda self, *a, **k: self.double()   # the reference calls .float() on inputs
            try:
                tr64 = ref_trainer.WorldModelTrainer(cfg_dict)
                tr64.train()
                tr64.preprocess.eval()
                tr64.model.load_state_dict(model.state_dict())
                tr64.double()
                for m in tr64.model.modules():
                    if isinstance(m, torch.nn.Dropout):
                        m.p = 0.0
                    if isinstance(m, torch.nn.MultiheadAttention):
                        m.dropout = 0.0
                if args.fp64_checkpoint:
                    # float64 convolutions on CPU take torch's im2col path, whose column buffer is
                    # taps * Cin * voxels * 8 B PER FRAME (8 GB for the 16->8 Conv3d at 192x192x64): feed the convolution
                    # modules one frame at a time (a convolution is independent per batch element)
                    def per_frame(fwd):
                        return lambda x: torch.cat([fwd(x[i:i + 1]) for i in range(x.shape[0])], 0) if x.shape[0] > 1 else fwd(x)
                    for m in tr64.model.modules():
                        if isinstance(m, (torch.nn.Conv3d, torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                            m.forward = per_frame(m.forward)
                    # activation memory: the decoders hold most of it (float64: ~6 GB per frame).  Recompute them in
                    # backward instead (torch.utils.checkpoint around the REAL modules' forward; deterministic CPU
                    # arithmetic, so the gradients are the same numbers)
                    from torch.utils.checkpoint import checkpoint
                    vd = tr64.model.voxel_decoder
                    mods = [tr64.model.rgb_decoder, tr64.model.lidar_re, tr64.model.encoder, tr64.model.range_view_encoder,
                            tr64.model.feat_decoder, tr64.model.range_view_decoder,
                            vd.first_conv, *vd.middle_conv, vd.conv1, vd.conv2, vd.conv3]
                    for mod in mods:     # (the voxel decoder block by block: its top level alone is ~2.6 GB per frame)
                        mod.forward = (lambda *a, _f=mod.forward: checkpoint(_f, *a, use_reentrant=False))
                b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in raw.items()}
                with NoisePatch(eps.double(), coin):
                    out64, _ = tr64.forward(b64)
                l64 = tr64.compute_loss(b64, out64)
                tot64 = tr64.loss_reducing(l64)
                tot64.backward()
            finally:
                torch.set_default_dtype(torch.float32)
                torch.Tensor.float = _tensor_float
            p64 = dict(tr64.model.named_parameters())
            rec['total_fp64'] = float(tot64.item())
            rec['losses_fp64'] = {k: float(v.item()) for k, v in l64.items()}
            gn64, noise = {}, {}
            for n, p in model.named_parameters():
                if p.grad is None:
                    gn64[n] = None
                    continue
                g64 = p64[n].grad
                gn64[n] = float(g64.pow(2).sum().sqrt())
                noise[n] = float((p.grad.double() - g64).pow(2).sum().sqrt())   # L2 of the fp32 reference's own error
            rec['grad_l2_fp64'] = gn64
            rec['grad_l2_ref32_err'] = noise
            for key in [k for k in samples if k.startswith('grad.')]:
                n = key[5:]
                g64 = p64[n].grad.detach().contiguous().view(-1)
                stride = max(1, g64.numel() // 1024)
                samples['grad64.' + n] = g64[::stride][:1024].clone().numpy()
            import resource
            print(f'  peak RSS so far {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GB')
            print(f'  fp64 reference fwd+bwd {time.time() - t0:.1f}s total={tot64.item():.9f} '
                  f'(fp32 total {rec["total"]:.9f})')
            del tr64, out64, p64

        rec['lr'] = [g['lr'] for g in opt.param_groups]
        opt.step()
        sched.step()
        ps = {}
        for n, p in model.named_parameters():
            d = p.detach().double()
            ps[n] = [float(d.sum()), float(d.abs().sum())]
        rec['param_checksums_after_step'] = ps
        if oracle_model is not None:
            pdev = max(float((p - dict(model.named_parameters())[n]).abs().max())
                       for n, p in oracle_model.named_parameters())
            rec['oracle_vs_reference']['max_abs_param_dev_after_step'] = pdev
            print(f'  max abs param dev after step: {pdev:.3e}')
        fixture['steps'].append(rec)

    with open(os.path.join(REPO, 'tests', 'golden', f'base1d_{tag}.json'), 'w') as f:
        json.dump(fixture, f)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', f'base1d_{tag}_samples.npz'), **samples)
    print('wrote fixtures for', tag)


if __name__ == '__main__':
    main()
