"""Host issue time of one training step: time until the Python side has queued everything vs. time until the GPU is done.
    python tools/host_issue_time.py [batch] [seq]   (batch 1, seq 1 makes the GPU work tiny: the total is then the host cost)"""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = base_1d_cfg(RECEPTIVE_FIELD=max(S - 4, 1), FUTURE_HORIZON=S - max(S - 4, 1), BATCHSIZE=B, STEPS=100000)
torch.manual_seed(1234)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev); tr.train(); tr.preprocess.augment = False
opts, scheds = tr.configure_optimizers(); opt, sched = opts[0], scheds[0]['scheduler']
batches = [make_batch(B, S, seed=1234 + k, device=dev) for k in range(2)]
def step(i):
    opt.zero_grad(); loss = tr.training_step(dict(batches[i % 2]), i); loss.backward(); opt.step(); sched.step(); return loss
for i in range(2): step(i)
torch.cuda.synchronize()
for i in range(4):
    c0 = time.process_time(); t0 = time.perf_counter(); step(2 + i); t1 = time.perf_counter(); c1 = time.process_time()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    # process CPU time (all threads: main + autograd engine) = what a rank costs its host; wall issue time also counts the
    # time the launching thread spends blocked on a full HIP queue
    print(f'host issue {1e3*(t1-t0):.1f} ms wall, {1e3*(c1-c0):.1f} ms CPU (all threads), gpu tail {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f} ms')
