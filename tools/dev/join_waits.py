"""How long does the main stream wait at each branch join of the forward pass (unprofiled)?  Monkeypatches ops.branch.join to
record one event on the side stream (end of the branch) and one on the main stream (just before the wait); prints
side_end - main_ready per branch, mean over steps (positive: the main stream waited that long)."""
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
real_join = ops.branch.join


def join(self):
    if self.on and not self.joined:
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(self.side)
        b.record(torch.cuda.current_stream(self.device))
        rec.append((self.name, a, b))
    return real_join(self)


ops.branch.join = join


def step(i):
    opt.zero_grad(); loss = tr.training_step(dict(batches[i % 2]), i); loss.backward(); opt.step(); sched.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
rec.clear()
for i in range(10):
    step(4 + i)
torch.cuda.synchronize()
acc = {}
for name, a, b in rec:
    acc.setdefault(name, []).append(b.elapsed_time(a))
for name, v in acc.items():
    print(f'{name:16s} side stream ends {sum(v) / len(v):+7.2f} ms after the main stream is ready to join (min {min(v):+.2f}, max {max(v):+.2f}, {len(v)} joins)')
