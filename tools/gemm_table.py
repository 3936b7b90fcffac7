"""GEMM shapes of one training step with per-shape GPU time (HIP events around every muvo_gemm call).
    python tools/gemm_table.py"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer
dev = torch.device('cuda:0')
B, S = 2, 10
cfg = base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=B, STEPS=100000)
torch.manual_seed(1234)
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev); tr.train(); tr.preprocess.augment = False
opts, scheds = tr.configure_optimizers(); opt, sched = opts[0], scheds[0]['scheduler']
batches = [make_batch(B, S, seed=1234 + k, device=dev) for k in range(2)]
def step(i):
    opt.zero_grad(); loss = tr.training_step(dict(batches[i % 2]), i); loss.backward(); opt.step(); sched.step(); return loss
for i in range(2): step(i)
torch.cuda.synchronize()
rec = []
strides = set()
_gemm = ops.gemm
def logged(A, Bm, Cout, M, N, K, sam, sak, sbk, sbn, scm, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _gemm(A, Bm, Cout, M, N, K, sam, sak, sbk, sbn, scm, *a, **kw); e1.record()
    rec.append(((M, N, K, kw.get('B1', 1) * kw.get('B2', 1), int(sak == 1), int(sbn == 1), kw.get('mode', 0)), e0, e1))
    if M <= 32 and os.environ.get('GEMM_TABLE_STRIDES'):
        strides.add((M, N, K, sam, sak, sbk, sbn, scm, (A.data_ptr() + 4 * kw.get('a_off', 0)) % 16, (Bm.data_ptr() + 4 * kw.get('b_off', 0)) % 16,
                     kw.get('bias') is not None, kw.get('act', 0)))
ops.gemm = logged
STEPS = 3
for i in range(STEPS): step(2 + i)
torch.cuda.synchronize()
tab = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    tab[key][0] += 1; tab[key][1] += e0.elapsed_time(e1)
rows = sorted(tab.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in tab.values()) / STEPS
print(f'# {len(rec) // STEPS} gemm calls/step, {tot:.2f} ms/step (event time, includes launch gaps)')
for t in sorted(strides):
    print('# skinny M N K sam sak sbk sbn scm A%16 B%16 bias act:', t)
print('   M      N      K  batch kA nB mode  calls/step  ms/step   us/call  TFLOP/s')
for (M, N, K, b, ka, nb, mode), (n, ms) in rows:
    fl = 2.0 * M * N * K * b
    print(f'{M:6d} {N:6d} {K:6d} {b:5d}  {ka}  {nb}  {mode}   {n / STEPS:8.1f} {ms / STEPS:8.3f} {1e3 * ms / n:9.1f} {fl * n / (ms * 1e-3) * 1e-12:8.1f}')
