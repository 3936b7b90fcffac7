"""Do an MFMA-bound convolution kernel (one 147-KB-LDS workgroup per CU) and HBM-bound elementwise kernels share the chip when
they sit on two HIP streams?  Serial vs concurrent wall time of the same launches (decides whether the weight-gradient kernels
should move to a side stream).  python tools/overlap_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muvo_amd import nn as hnn, ops

dev = torch.device('cuda', 0)
torch.manual_seed(0)
with torch.device(dev):
    big = hnn.ConvTranspose2d(128, 64, 6, 2, 2)
    small = hnn.Conv2d(64, 64, 3, 1, 1)
xb = torch.randn(20, 128, 160, 416, device=dev)
xs = torch.randn(20, 64, 80, 208, device=dev)
e = torch.randn(20, 64, 320, 832, device=dev)        # 1.36 GB
e2 = torch.empty_like(e)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def mfma(n, which):
    with torch.no_grad():
        for _ in range(n):
            (big(xb) if which == 'big' else small(xs))


def hbm(n):
    for _ in range(n):
        torch.add(e, 1.0, out=e2)


def timed(fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def both(n1, which, n2, concurrent):
    def run():
        if not concurrent:
            mfma(n1, which); hbm(n2)
            return
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            mfma(n1, which)
        with torch.cuda.stream(s2):
            hbm(n2)
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run


def two_mfma(n, concurrent):
    def run():
        if not concurrent:
            mfma(n, 'small'); mfma(n, 'small')
            return
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            mfma(n, 'small')
        with torch.cuda.stream(s2):
            mfma(n, 'small')
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run


for _ in range(2):
    mfma(2, 'big'); mfma(2, 'small'); hbm(2)
print('big conv x4 alone      %.3f ms' % timed(lambda: mfma(4, 'big')))
print('small conv x40 alone   %.3f ms' % timed(lambda: mfma(40, 'small')))
print('hbm add x16 alone      %.3f ms' % timed(lambda: hbm(16)))
for which, n1, n2 in (('big', 4, 16), ('small', 40, 16)):
    for conc in (False, True, False, True):
        print('%s conv x%d + add x%d, %s: %.3f ms' % (which, n1, n2, 'two streams' if conc else 'one stream ', timed(both(n1, which, n2, conc))))
for conc in (False, True, False, True):
    print('small conv x40 twice, %s: %.3f ms' % ('two streams' if conc else 'one stream ', timed(two_mfma(40, conc))))
