"""GPU box: achieved HBM bandwidth of the memory-bound kernels at their largest base_1d shapes (HIP events, 20 repetitions).
   python tools/hbm_bench.py [name ...]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from muvo_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
L = ops.lib()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def report(name, sec, nbytes):
    print(f'{name:44s} {sec * 1e6:9.1f} us  {nbytes / 1e9:7.3f} GB  {nbytes / sec / 1e12:6.2f} TB/s', flush=True)


def split(n, c, s, with_act=False):
    x = torch.randn(n, c, s, device=dev)
    y = torch.randn(n, c, s, device=dev) if with_act else None
    db = torch.zeros(c, device=dev) if with_act else None
    ws = torch.empty((L.muvo_split_planes_bytes(n, c, C.c_int64(s)) + 3) // 4, device=dev)
    f = lambda: ops._ck(L.muvo_split_planes(ops._f(x), ops._p(ws), n, c, C.c_int64(s), ops._f(y), ops.ACT_ELU if with_act else 0,
                                           C.c_float(0.0), ops._f(db), ops._st()))
    report(f'split_planes n{n} c{c} s{s}{" +act" if with_act else ""}', timeit(f), x.numel() * (12 if with_act else 8))


def main():
    want = set(sys.argv[1:])
    def on(k):
        return not want or k in want
    if on('split'):
        for (n, c, s) in [(20, 128, 160 * 416), (20, 256, 80 * 208), (20, 512, 40 * 104), (20, 64, 320 * 832), (20, 64, 80 * 208),
                          (20, 128, 40 * 104), (20, 512, 10 * 26), (20, 32, 96 * 96 * 32)]:
            split(n, c, s)
        split(20, 128, 160 * 416, True)
        split(20, 64, 320 * 832, True)
    if on('maxpool'):
        x = torch.randn(20, 64, 160, 416, device=dev, requires_grad=True)
        y = ops.max_pool2d(x, 3, 2, 1)
        g = torch.randn_like(y)
        report('maxpool3s2 fwd 20x64x160x416', timeit(lambda: ops.max_pool2d(x.detach(), 3, 2, 1)), x.numel() * 4 + y.numel() * 8)
        report('maxpool3s2 bwd', timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True)), x.numel() * 4 + y.numel() * 8)   # (autograd.grad: no AccumulateGrad add pass in the timing)
    if on('upsample'):
        x = torch.randn(20, 16, 96, 96, 32, device=dev, requires_grad=True)
        y = ops.upsample3d_x2(x)
        g = torch.randn_like(y)
        report('upsample3d fwd 20x16x96x96x32', timeit(lambda: ops.upsample3d_x2(x.detach())), x.numel() * 4 + y.numel() * 4)
        report('upsample3d bwd', timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True)), x.numel() * 4 + y.numel() * 4)
    if on('voxloss'):
        lg = torch.randn(2, 10, 2, 192, 192, 64, device=dev, requires_grad=True)
        lab = (torch.rand(2, 10, 1, 192, 192, 64, device=dev) < 0.1).to(torch.uint8)
        out = ops.voxel_losses(lg, lab, 0.1)
        report('voxel_losses fwd 20x2x192x192x64', timeit(lambda: ops.voxel_losses(lg.detach(), lab, 0.1)), lg.numel() * 4 + lab.numel())
        report('voxel_losses bwd', timeit(lambda: out.sum().backward(retain_graph=True)), lg.numel() * 8 + lab.numel())
    if on('bn'):
        from muvo_amd import nn as hnn
        with torch.device(dev):
            bn = hnn.BatchNorm2d(64)
        x = torch.randn(20, 64, 160, 416, device=dev, requires_grad=True)
        y = bn(x, relu=True)
        g = torch.randn_like(y)
        report('bn+relu fwd 20x64x160x416 (2 reads 1 write)', timeit(lambda: bn(x.detach(), relu=True)), x.numel() * 12)
        report('bn+relu bwd (stats: x,dy,y; apply: x,dy,y->dx)', timeit(lambda: y.backward(g, retain_graph=True)), x.numel() * 28)
    if on('adain'):
        x = torch.randn(20, 8, 192, 192, 64, device=dev, requires_grad=True)
        st = torch.randn(20, 16, device=dev, requires_grad=True)
        y = ops.adain(x, st, 1e-8, 20, ops.ACT_LEAKY, 0.2)
        g = torch.randn_like(y)
        report('adain fwd 20x8x192x192x64 (2 reads 1 write)', timeit(lambda: ops.adain(x.detach(), st.detach(), 1e-8, 20)), x.numel() * 12)
        report('adain bwd (stats: x,dy; apply: x,dy->dx)', timeit(lambda: y.backward(g, retain_graph=True)), x.numel() * 20)
    if on('spatial'):
        p = torch.randn(2, 10, 3, 320, 832, device=dev, requires_grad=True)
        t = torch.rand(2, 10, 3, 320, 832, device=dev)
        o = ops.spatial_losses(p, t, [(0, 3, 1, 0.1)])
        report('spatial L1 fwd 20x3x320x832', timeit(lambda: ops.spatial_losses(p.detach(), t, [(0, 3, 1, 0.1)])), p.numel() * 8)
        report('spatial L1 bwd', timeit(lambda: o.sum().backward(retain_graph=True)), p.numel() * 12)


if __name__ == '__main__':
    main()
