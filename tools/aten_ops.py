"""Which ATen operators (kernels that are not ours) does one training step launch, and with which shapes?
   python tools/aten_ops.py   (GPU box; prints the most frequent (operator, input shapes) pairs of one step)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
rows = collections.Counter()
stacks = {}
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::'):
        continue
    if e.name in ('aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::copy_', 'aten::cat', 'aten::mul', 'aten::sum',
                  'aten::contiguous', 'aten::clone', 'aten::zeros_like', 'aten::div', 'aten::sub', 'aten::mean', 'aten::stack', 'aten::index',
                  'aten::_to_copy', 'aten::neg', 'aten::exp', 'aten::where', 'aten::masked_fill_', 'aten::gather', 'aten::index_select'):
        key = (e.name, str(e.input_shapes)[:90])
        rows[key] += 1
        if key not in stacks and e.stack:
            stacks[key] = [s for s in e.stack if 'muvo_amd' in s][:3]
for (name, shp), n in rows.most_common(45):
    print(f'{n:4d}  {name:18s} {shp}')
    for s in stacks.get((name, shp), []):
        print('        ', s[-110:])
