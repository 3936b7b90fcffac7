"""Run-to-run reproducibility of the parameter gradients of one training step (same batch, same noise): float atomics
change the last digits, anything larger points at a race.  python tools/grad_repeat.py [runs]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch, make_noise
from muvo_amd.trainer import WorldModelTrainer
from muvo_amd.utils import detinit

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
b, s, seed = 1, 2, 1234
cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
tr.train()
tr.preprocess.augment = False
detinit.fill_state_dict_(tr.model)
for layer in tr.model.transformer_encoder.layers:
    layer.p = 0.0
eps, use_prior = make_noise(b, s, seed=seed)
eps = eps.to(dev)
grads = []
outs = []
for r in range(runs):
    batch = make_batch(b, s, seed=seed, device=dev)
    for p in tr.model.parameters():
        p.grad = None
    tr.zero_grad()
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps, use_prior=use_prior)
    tr.loss_reducing(losses).backward()
    outs.append({k: v.detach().clone() for k, v in output.items() if torch.is_tensor(v)})
    print('run', r, 'total', repr(tr.loss_reducing(losses).item()))
    grads.append({n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None})
worst = []
for n in grads[0]:
    ref = grads[0][n]
    scale = ref.abs().max().item() + 1e-30
    dev_ = max((g[n] - ref).abs().max().item() for g in grads[1:]) / scale
    worst.append((dev_, n, scale))
worst.sort(reverse=True)
for d, n, sc in worst[:12]:
    print(f'{d:10.3e}  (max |g| {sc:.3e})  {n}')

print('forward outputs, run-to-run max deviation / max |x|:')
for k in outs[0]:
    ref = outs[0][k].float()
    d = max((o[k].float() - ref).abs().max().item() for o in outs[1:]) / (ref.abs().max().item() + 1e-30)
    print(f'  {d:10.3e}  {k}')
import collections
by = collections.OrderedDict()
for d, n, sc in worst:
    top = '.'.join(n.split('.')[:2])
    by[top] = max(by.get(top, 0.0), d)
print('per module (max over its tensors):')
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f'  {v:10.3e}  {k}')

print('rgb_decoder / lidar_re tensors:')
for d, n, sc in sorted(worst, key=lambda t: t[1]):
    if n.startswith('rgb_decoder') or n.startswith('lidar_re.pre'):
        print(f'  {d:10.3e}  (max |g| {sc:.3e})  {n}')
