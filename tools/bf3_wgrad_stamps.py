"""Per-workgroup phases of conv_bf3_wgrad_pp_kernel (eight-wave weight-gradient tiles).
   tools/ab_build.sh "-DMUVO_BF3_STAMPS=3" st3 -- python tools/bf3_wgrad_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from muvo_amd import nn as hnn, ops

dev = torch.device('cuda', 0)
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
L = ops.lib()
L.muvo_debug_bf3_stamps.argtypes = [C.c_void_p, C.c_int]
LAYERS = [('convT 128->64 in 160x416', lambda: hnn.ConvTranspose2d(128, 64, 6, 2, 2), (20, 128, 160, 416)),
          ('convT 512->256 in 40x104', lambda: hnn.ConvTranspose2d(512, 256, 6, 2, 2), (20, 512, 40, 104)),
          ('conv 128->128 3x3 in 40x104', lambda: hnn.Conv2d(128, 128, 3, 1, 1), (20, 128, 40, 104)),
          ('conv 256->256 3x3 in 20x52', lambda: hnn.Conv2d(256, 256, 3, 1, 1), (20, 256, 20, 52)),
          ('conv 512->512 3x3 in 10x26', lambda: hnn.Conv2d(512, 512, 3, 1, 1), (20, 512, 10, 26)),
          ('convT 512->512 k5 in 10x26', lambda: hnn.ConvTranspose2d(512, 512, 5, 2, 2, 1), (20, 512, 10, 26)),
          ('convT 512->512 k5 in 5x13', lambda: hnn.ConvTranspose2d(512, 512, 5, 2, 2, 1), (20, 512, 5, 13)),
          ('convT 512->512 k6 in 4x64', lambda: hnn.ConvTranspose2d(512, 512, 6, 2, 2), (20, 512, 4, 64))]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if sys.argv[1] in l[0]]
for name, make, shape in LAYERS:
    torch.manual_seed(0)
    with torch.device(dev):
        m = make()
    x = torch.randn(*shape, device=dev, requires_grad=True)
    y = m(x)
    g = torch.randn_like(y)
    for _ in range(2):
        m.weight.grad = None
        y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    assert L.muvo_debug_bf3_stamps_reset() == 0
    y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 16384, dtype=np.uint64)
    assert L.muvo_debug_bf3_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(16384, 8)
    st = st[st[:, 0] > 0]
    t = st[:, [0, 1, 2, 3, 4]].astype(np.int64)
    us = (t - t[:, 0].min()) / 100.0
    ns = st[:, 7].astype(np.float64)
    hw = st[:, 5]
    cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 13) & 0x7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    print(f'== {name}: {len(st)} workgroups recorded, K steps per workgroup median {np.median(ns):.0f}, span {us[:, 4].max():.1f} us, {len(np.unique(cu))} CUs')
    for lbl, d in (('setup', us[:, 1] - us[:, 0]), ('prologue', us[:, 2] - us[:, 1]), ('K loop', us[:, 3] - us[:, 2]),
                   ('epilogue issue', us[:, 4] - us[:, 3]), ('whole workgroup', us[:, 4] - us[:, 0])):
        print(f'   {lbl:16s} mean {d.mean():8.2f}  p10 {np.percentile(d, 10):8.2f}  median {np.median(d):8.2f}  p90 {np.percentile(d, 90):8.2f} us')
    print(f'   us per K step in the loop: {np.median((us[:, 3] - us[:, 2]) / np.maximum(ns, 1)):.3f}')
    gaps = []
    for c in np.unique(cu):
        sel = us[cu == c]
        sel = sel[np.argsort(sel[:, 0])]
        gaps += list(sel[1:, 0] - sel[:-1, 4])
    if gaps:
        gaps = np.array(gaps)
        print(f'   gap end -> next start on the same CU: median {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us')
