"""Single-layer benchmark of the conv family through the C ABI (GPU box): forward / dgrad / wgrad time and
algorithmic TFLOP/s for the decoder shapes that dominate the step.  Used for kernel tuning and as the command
under rocprofv3 --pmc.

    python tools/layer_bench.py [--mode f32|bf16x3] [--layers rgb3,rgb2,...] [--iters 10] [--what fwd,dgrad,wgrad]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd import nn as hnn  # noqa: E402
from muvo_amd import ops  # noqa: E402

LAYERS = {
    # name: (kind, cin, cout, k, stride, pad, in_sz)   at N = 20 frames (base_1d, batch 2 x seq 10)
    'rgb1': ('convT', 512, 256, 6, 2, 2, (40, 104)),
    'rgb2': ('convT', 256, 128, 6, 2, 2, (80, 208)),
    'rgb3': ('convT', 128, 64, 6, 2, 2, (160, 416)),
    'rgb0': ('convT', 512, 512, 6, 2, 2, (20, 52)),
    'lid3': ('convT', 128, 64, 6, 2, 2, (32, 512)),
    'res64': ('conv', 64, 64, 3, 1, 1, (80, 208)),
    'ds128': ('conv', 128, 384, 3, 1, 1, (40, 104)),
    'res64rv': ('conv', 64, 64, 3, 1, 1, (16, 256)),
    'ds64': ('conv', 64, 128, 3, 2, 1, (80, 208)),
    'res512': ('conv', 512, 512, 3, 1, 1, (10, 26)),
    'vox16': ('conv3d', 16, 8, 3, 1, 1, (192, 192, 64)),
    'vox8': ('conv3d', 8, 8, 3, 1, 1, (192, 192, 64)),
    'vox16b': ('conv3d', 16, 16, 3, 1, 1, (96, 96, 32)),
    'vox32': ('conv3d', 32, 16, 3, 1, 1, (96, 96, 32)),
    'vox64': ('conv3d', 64, 64, 3, 1, 1, (24, 24, 8)),
    'vox64s': ('conv3d', 64, 32, 3, 1, 1, (48, 48, 16)),
    'vox32s': ('conv3d', 32, 32, 3, 1, 1, (48, 48, 16)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='f32')
    ap.add_argument('--layers', default='rgb1,rgb2,rgb3')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--what', default='fwd,dgrad,wgrad')
    ap.add_argument('--n', type=int, default=20)
    ap.add_argument('--zeros', action='store_true', help='all-zero activations (power experiment: same instructions, no bit toggling)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    ops.set_conv_mode(ops.CONV_BF16X3 if args.mode == 'bf16x3' else ops.CONV_F32)
    what = args.what.split(',')
    for name in args.layers.split(','):
        kind, cin, cout, k, s, p, sz = LAYERS[name]
        torch.manual_seed(0)
        with torch.device(dev):
            if kind == 'convT':
                m = hnn.ConvTranspose2d(cin, cout, k, s, p)
            elif kind == 'conv3d':
                m = hnn.Conv3d(cin, cout, k, s, p)
            else:
                m = hnn.Conv2d(cin, cout, k, s, p)
        x = torch.randn(args.n, cin, *sz, device=dev, requires_grad=True)
        if args.zeros:
            x = torch.zeros_like(x).requires_grad_(True)
            with torch.no_grad():
                m.weight.zero_()
        y = m(x)
        g = torch.randn_like(y)
        m.weight.grad = torch.zeros_like(m.weight)
        m.bias.grad = torch.zeros_like(m.bias)
        taps = k ** len(sz)
        pix = math.prod(sz) if kind == 'convT' else math.prod(y.shape[2:])
        flop = 2.0 * args.n * cin * cout * taps * pix
        L = ops.lib()
        geom, packed = m.geom, m._packed
        in_sz = tuple(sz) if len(sz) == 3 else (1,) + tuple(sz)
        d, out_sz, ff, df = geom.plan(args.n, in_sz)
        y.backward(g)  # packs dgrad weights, warms everything
        ws = ops.scratch_zeroed('wgrad', ff, dev)
        import ctypes as C

        def run(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters
        xd, gd = x.detach(), g
        wb = geom.ws_bytes[(args.n, in_sz, ops._plan_epoch[0])]
        wsf = torch.empty(wb[0] // 4 + 1, device=dev) if wb[0] else None
        wsd = torch.empty(wb[1] // 4 + 1, device=dev) if wb[1] else None
        wsx = torch.empty(wb[2] // 4 + 1, device=dev) if wb[2] else None
        wsy = torch.empty(wb[3] // 4 + 1, device=dev) if wb[3] else None
        dx = torch.empty_like(xd)
        fns = {
            'fwd': lambda: ops._ck(L.muvo_conv_forward(C.byref(d), ops._f(xd), ops._f(packed.fwd), ops._f(m.bias), ops._f(y.detach()), 0, ops._fl(0.0), ops._p(wsf), ops._st())),
            'dgrad': lambda: ops._ck(L.muvo_conv_dgrad(C.byref(d), ops._f(gd), ops._f(packed.dgr), ops._f(dx), ops._p(wsd), 0, ops._st())),
            'wgrad': lambda: ops._ck(L.muvo_conv_wgrad(C.byref(d), ops._f(xd), ops._f(gd), ops._f(ws), ops._f(m.weight.grad), ops._f(m.bias.grad), ops._p(wsx), ops._p(wsy), 0, ops._st())),
        }
        for w in what:
            ms = run(fns[w])
            print(f'{name:7s} {args.mode:7s} {w:6s} {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
