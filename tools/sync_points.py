"""Which operations of one training step synchronise the host with the device?  torch.cuda.set_sync_debug_mode('warn') prints
a warning with a stack for every synchronising call (GPU box).   python tools/sync_points.py"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer

dev = torch.device('cuda', 0)
ops.set_conv_mode(ops.CONV_BF16X3)
cfg = base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=2, STEPS=100000)
torch.manual_seed(1234)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
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


for i in range(2):
    step(i)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step(2)
torch.cuda.set_sync_debug_mode('default')
print(f'{len(w)} synchronising calls in one step')
import collections
c = collections.Counter((str(x.filename)[-40:], x.lineno) for x in w)
for (f, l), n in c.most_common(20):
    print(f'  {n:3d} x {f}:{l}')
