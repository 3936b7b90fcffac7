"""Which Python call sites of one training step produce the ~380 fill / memset / copy / add launches of the kernel statistics?
Counts calls of torch.zeros / zeros_like / full / Tensor.zero_ / fill_ / copy_ / clone / contiguous (copying) / add on CUDA tensors by the
innermost muvo_amd (or bench / trainer) frame that made them.    python tools/dev/small_launch_census.py   (GPU box)"""
import collections
import os
import sys
import traceback
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
opts, scheds = tr.configure_optimizers()
opt, sched = opts[0], scheds[0]['scheduler']
batches = [make_batch(2, 10, seed=1234 + k, device=dev) for k in range(2)]


def step(i):
    opt.zero_grad()
    loss = tr.training_step(dict(batches[i % 2]), i)
    loss.backward()
    tr.on_after_backward()
    opt.step()
    sched.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
counts = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if '/muvo_amd/' in fr.filename or fr.filename.endswith('small_launch_census.py'):
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}'
    return '(torch internal / autograd engine)'


def wrap_fn(mod, name, is_cuda):
    orig = getattr(mod, name)

    def f(*a, **k):
        r = orig(*a, **k)
        if ON[0] and is_cuda(a, k, r):
            counts[(name, site())] += 1
        return r
    setattr(mod, name, f)


def out_cuda(a, k, r):
    return torch.is_tensor(r) and r.is_cuda


def self_cuda(a, k, r):
    return torch.is_tensor(a[0]) and a[0].is_cuda


for n in ('zeros', 'zeros_like', 'full', 'full_like', 'ones', 'tensor'):
    wrap_fn(torch, n, out_cuda)
for n in ('zero_', 'fill_', 'copy_', 'clone', 'add_', 'add', '__add__', '__iadd__', 'to', 'float'):
    wrap_fn(torch.Tensor, n, self_cuda)
_c = torch.Tensor.contiguous


def contig(self, *a, **k):
    if ON[0] and self.is_cuda and not self.is_contiguous():
        counts[('contiguous(copy)', site())] += 1
    return _c(self, *a, **k)


torch.Tensor.contiguous = contig
ON[0] = True
step(3)
torch.cuda.synchronize()
ON[0] = False
tot = collections.Counter()
for (name, s), n in counts.items():
    tot[name] += n
print('calls per step by operation:', dict(tot))
for (name, s), n in counts.most_common(70):
    print(f'{n:4d}  {name:18s} {s}')
