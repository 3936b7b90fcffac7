"""Deterministic mode: two runs of one training step (same batch, same noise) must be BIT-identical: prints every tensor that
differs.  python tools/det_repeat.py [f32|policy] [b] [s]"""
import sys
import torch
sys.path.insert(0, '.')
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch, make_noise
from muvo_amd.trainer import WorldModelTrainer
from muvo_amd.utils import detinit

mode = sys.argv[1] if len(sys.argv) > 1 else 'policy'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
ops.set_conv_mode(ops.CONV_F32 if mode == 'f32' else ops.CONV_BF16X3, min_gflop=-1.0)
import os
DET = os.environ.get('DET', '1') != '0'
ops.set_deterministic(DET)
tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000).convert_to_dict(), device=dev)
tr.train()
tr.preprocess.augment = False
detinit.fill_state_dict_(tr.model)
opt = tr.configure_optimizers()[0][0]
eps, use_prior = make_noise(b, s, seed=1234)
eps = eps.to(dev)
state = {k: v.clone() for k, v in tr.model.state_dict().items()}
runs = []
for r in range(3):
    tr.model.load_state_dict(state)
    tr.model._step_seed = 0                      # same dropout masks
    opt.zero_grad()
    batch = make_batch(b, s, seed=1234, device=dev)
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps, use_prior=use_prior)
    torch.cuda.synchronize()
    pre = {'pre.' + k: v.detach().clone() for k, v in output.items() if torch.is_tensor(v)}      # outputs BEFORE backward
    torch.cuda.synchronize()
    tr.loss_reducing(losses).backward()
    torch.cuda.synchronize()
    rec = {'loss.' + k: v.detach().clone() for k, v in losses.items()}
    rec.update(pre)
    for k, v in output.items():
        if torch.is_tensor(v) and not torch.equal(v, pre['pre.' + k]):
            print(f'run {r}: output {k} CHANGED during backward: {(v - pre["pre." + k]).abs().gt(0).sum().item()} elements')
    rec.update({'out.' + k: v.detach().clone() for k, v in output.items() if torch.is_tensor(v)})
    rec.update({'grad.' + n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None})
    rec.update({'buf.' + n: v.detach().clone() for n, v in tr.model.named_buffers()})
    runs.append(rec)
for a, b_ in ((0, 1), (1, 2), (0, 2)):
    d = {k: (runs[a][k].double() - runs[b_][k].double()).abs().max().item() / (runs[a][k].double().abs().max().item() + 1e-300)
         for k in runs[0] if k.startswith('out.voxel')}
    print(f'runs {a} vs {b_}:', {k: f'{v:.2e}' for k, v in d.items()})
bad = []
for k in runs[0]:
    for r in runs[1:]:
        if not torch.equal(runs[0][k], r[k]):
            d = (runs[0][k].double() - r[k].double()).abs().max().item() / (runs[0][k].double().abs().max().item() + 1e-300)
            bad.append((k, d))
            break
print(f'{mode} b{b}s{s}: {len(runs[0])} tensors compared over 3 runs, {len(bad)} differ')
for k, _ in bad[:4]:
    if runs[0][k].numel() > 1:
        for r_i, r in enumerate(runs[1:], 1):
            d = (runs[0][k] - r[k]).abs()
            nz = torch.nonzero(d.reshape(-1)).reshape(-1)
            if nz.numel():
                idx = nz[:6].tolist()
                print(f'   {k} run0 vs run{r_i}: {nz.numel()} of {d.numel()} elements differ, max {d.max().item():.3e}; shape {tuple(d.shape)}; first flat idx {idx}; '
                      f'unravel {[tuple(int(v) for v in torch.unravel_index(torch.tensor(i), d.shape)) for i in idx[:3]]}')
                a, b_ = runs[0][k].reshape(-1), r[k].reshape(-1)
                print('      run0 values', [round(float(a[i]), 5) for i in nz[:8].tolist()], ' other', [round(float(b_[i]), 5) for i in nz[:8].tolist()],
                      ' last idx', nz[-3:].tolist(), ' gaps', sorted(set((nz[1:] - nz[:-1]).tolist()))[:6])
thr = 0.0 if DET else 1e-3
shown = [(k, d) for k, d in bad if d > thr]
print(f'   ({len(shown)} above {thr})')
for k, d in shown[:60]:
    print(f'   {d:10.3e}  {k}')
