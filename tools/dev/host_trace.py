"""Where does the host block while it issues a training step?  Wraps every entry of the C library with a timer: prints the calls
that took > 0.3 ms on the host (a launch normally costs 5-20 us), with the stream they were issued on and the time since the
step began, for one steady-state step.   python tools/dev/host_trace.py"""
import sys, time, threading, torch
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
L = ops.lib()
log = []
T0 = [0.0]


_real_backward = torch.Tensor.backward


class Wrapped:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        t1 = time.perf_counter()
        log.append((t - T0[0], t1 - t, self.name, torch.cuda.current_stream().cuda_stream, threading.get_ident()))
        return r

    def __getattr__(self, k):
        return getattr(self.fn, k)


class Lib:
    def __init__(self, real):
        object.__setattr__(self, '_real', real)
        object.__setattr__(self, '_cache', {})

    def __getattr__(self, k):
        c = self._cache
        if k not in c:
            f = getattr(self._real, k)
            c[k] = Wrapped(k, f) if k.startswith('muvo_') and k not in ('muvo_last_error',) else f
        return c[k]


ops._lib = Lib(L)


def step(i):
    opt.zero_grad(); loss = tr.training_step(dict(batches[i % 2]), i)
    tf = time.perf_counter() - T0[0]
    loss.backward()
    tb = time.perf_counter() - T0[0]
    opt.step(); sched.step()
    return tf, tb


for i in range(3):
    step(i)
torch.cuda.synchronize()
NSTEADY = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # steps issued back to back before the traced one (steady state)
T0[0] = time.perf_counter()
for i in range(NSTEADY):
    step(10 + i)
log.clear()
T0[0] = time.perf_counter()
tf, tb = step(3)
te = time.perf_counter() - T0[0]
torch.cuda.synchronize()
tg = time.perf_counter() - T0[0]
print(f'forward issued at {1e3 * tf:.1f} ms, backward issued at {1e3 * tb:.1f} ms, step issued at {1e3 * te:.1f} ms, GPU done at {1e3 * tg:.1f} ms; '
      f'{len(log)} library calls, {1e3 * sum(l[1] for l in log):.1f} ms inside them')
streams = {}
for l in log:
    streams.setdefault(l[3], len(streams))
print('calls > 0.3 ms (time since step start ms, duration ms, entry, stream #, thread):')
for l in log:
    if l[1] > 3e-4:
        print(f'  {1e3 * l[0]:7.2f}  {1e3 * l[1]:6.2f}  {l[2]:34s} s{streams[l[3]]}  t{l[4] % 1000}')
# gaps between consecutive library calls (host busy elsewhere: Python, torch ops, allocator)
gaps = sorted(((b[0] - (a[0] + a[1]), a, b) for a, b in zip(log[:-1], log[1:])), key=lambda g: -g[0])[:12]
print('longest host gaps BETWEEN library calls (ms, after entry -> before entry, at ms):')
for g, a, b in gaps:
    print(f'  {1e3 * g:6.2f}  {a[2]} -> {b[2]}  at {1e3 * a[0]:.1f}')
