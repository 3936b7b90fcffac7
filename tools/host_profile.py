"""cProfile of the host side of one training step (GPU box): where the ~140 ms of Python/ctypes launch time per step goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd.config import base_1d_cfg  # noqa: E402
from muvo_amd.data.synthetic import make_batch  # noqa: E402
from muvo_amd.trainer import WorldModelTrainer  # noqa: E402

dev = torch.device('cuda:0')
cfg = base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=2, STEPS=100000)
torch.manual_seed(1234)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
tr.train()
tr.preprocess.augment = False
opts, scheds = tr.configure_optimizers()
opt, sched = opts[0], scheds[0]['scheduler']
batches = [make_batch(2, 10, seed=1234 + k, device=dev) for k in range(2)]


def step(i):
    opt.zero_grad()
    loss = tr.training_step(dict(batches[i % 2]), i)
    loss.backward()
    opt.step()
    sched.step()


for i in range(2):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    step(2 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
