"""dev: which elements of which parameters explain a checksum deviation of tests/test_dp_gpu.py::test_two_rank_step_matches_reference
(policy mode)?  Prints, per deviating parameter, the deviation of the sum in units of 2 lr and the smallest |gradient| / largest."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from muvo_amd import ops
from muvo_amd.data.synthetic import make_batch, make_noise
import test_dp_gpu as T
dev = torch.device('cuda:0')
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=-1.0)
fx = json.load(open(os.path.join(T.GOLD, 'base1d_dp2_b1s2.json')))
world, b, s, seed = fx['world'], fx['b'], fx['s'], fx['seed']
tr = T._trainer(dev, s)
opts, scheds = tr.configure_optimizers()
opt = opts[0]
opt.grad_scale = 1.0 / world
g = fx['steps'][0]
opt.zero_grad()
for rank in range(world):
    eps, use_prior = make_noise(b, s, seed=seed + 10 * rank)
    batch = make_batch(b, s, seed=seed + 10 * rank, device=dev)
    tr.training_step(batch, 0, noise=eps.to(dev), use_prior=use_prior).backward()
params = dict(tr.model.named_parameters())
before = {n: p.detach().double().clone() for n, p in params.items()}
grads = {n: (p.grad.detach().double().clone() if p.grad is not None else None) for n, p in params.items()}
lr = [pg['lr'] for pg in opt.param_groups]
opt.step()
nbad = 0
for n, (s_ref, a_ref) in g['param_checksums_after_step'].items():
    d = params[n].detach().double()
    if T._rel(d.abs().sum().item(), a_ref) > 1e-5 or abs(d.sum().item() - s_ref) > 1e-5 * max(a_ref, 1.0):
        nbad += 1
        gr = grads[n].abs().flatten()
        srt = gr.sort().values
        upd = (d - before[n]).flatten()
        print(n, 'numel', d.numel(), 'sum dev', d.sum().item() - s_ref, '= %.2f x 2lr' % ((d.sum().item() - s_ref) / (2 * lr[0])),
              'abs-sum rel dev %.2e' % T._rel(d.abs().sum().item(), a_ref), '| |g| min %.2e median %.2e max %.2e' % (srt[0].item(), srt[len(srt) // 2].item(), srt[-1].item()),
              '| |update| min %.2e max %.2e, lr %s' % (upd.abs().min().item(), upd.abs().max().item(), lr))
print('deviating parameters:', nbad)
