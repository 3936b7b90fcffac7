"""Which parameters still receive their gradient through an autograd AccumulateGrad node (instead of a kernel writing into the
flat gradient buffer), and on which stream does that node run?  bench.py's stderr carries autograd's warning "The AccumulateGrad
node's stream does not match the stream of the node that produced the incoming gradient".
   python tools/dev/accgrad_params.py   (GPU box)"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import muvo_amd  # noqa
import torch
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer

dev = torch.device('cuda', 0)
tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=2, STEPS=100000).convert_to_dict(), device=dev)
tr.train()
opt = tr.configure_optimizers()[0][0]
hits = {}
main = torch.cuda.current_stream(dev)
for n, p in tr.model.named_parameters():
    def hook(param, n=n):
        cur = torch.cuda.current_stream(dev)
        hits.setdefault(n, []).append('main' if cur == main else hex(cur.cuda_stream))
    p.register_post_accumulate_grad_hook(hook)
batches = [make_batch(2, 10, seed=1234 + k, device=dev) for k in range(2)]
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    for i in range(3):
        opt.zero_grad()
        loss = tr.training_step(dict(batches[i % 2]), i)
        loss.backward()
        tr.on_after_backward()
        opt.step()
    torch.cuda.synchronize()
print('side streams:', {k: hex(v.cuda_stream) for k, v in ops._side_streams.items()})
print(f'{len(hits)} parameters went through AccumulateGrad in 3 steps:')
for n, s in hits.items():
    p = dict(tr.model.named_parameters())[n]
    print(f'  {n:60s} {tuple(p.shape)}  streams {s}  grad is flat view: {p.grad is getattr(p, "_muvo_flat_grad", None)}')
for x in w:
    if 'AccumulateGrad' in str(x.message):
        print('WARNING raised:', str(x.message)[:200])
