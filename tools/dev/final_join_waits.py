"""How long does the main stream wait for each side stream when backward ends (the engine's final join / optimizer.step)?
Monkeypatches ops.join_side_streams: one event per side stream (its last queued work) and one on the current stream before the
wait; prints side_end - main_ready per stream (positive: the main stream waits that long before AdamW can start)."""
import sys, torch
sys.path.insert(0, '.')
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer

dev = torch.device('cuda:0')
cfg = base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=2, STEPS=100000)
torch.manual_seed(1234)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev); tr.train()
opts, scheds = tr.configure_optimizers(); opt, sched = opts[0], scheds[0]['scheduler']
batches = [make_batch(2, 10, seed=1234 + k, device=dev) for k in range(2)]
rec = []
real = ops.join_side_streams
tag = ['?']


def join(device=None, into=None):
    if ops._side_streams and into is None:
        b = torch.cuda.Event(enable_timing=True); b.record(torch.cuda.current_stream())
        for (name, idx), st in ops._side_streams.items():
            a = torch.cuda.Event(enable_timing=True); a.record(st)
            rec.append((tag[0], name, a, b))
    return real(device, into)


ops.join_side_streams = join
import muvo_amd.optim as optim_mod


def step(i):
    opt.zero_grad(); loss = tr.training_step(dict(batches[i % 2]), i)
    tag[0] = 'end of backward'
    loss.backward()
    tag[0] = 'optimizer.step'
    opt.step(); sched.step()
    tag[0] = 'other'


for i in range(4):
    step(i)
torch.cuda.synchronize(); rec.clear()
for i in range(10):
    step(4 + i)
torch.cuda.synchronize()
acc = {}
for t, name, a, b in rec:
    acc.setdefault((t, name), []).append(b.elapsed_time(a))
for (t, name), v in sorted(acc.items()):
    print(f'{t:16s} {name:4s} ends {sum(v) / len(v):+7.2f} ms after the main stream reaches the join (min {min(v):+.2f}, max {max(v):+.2f}, n={len(v)})')
