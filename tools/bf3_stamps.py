"""Where does a workgroup of conv_bf3_kernel (eight-wave ping-pong tiles, one 147-KB workgroup per CU) spend its time?
Needs the diagnostic build:  tools/ab_build.sh "-DMUVO_BF3_STAMPS=1" st -- python tools/bf3_stamps.py
Per layer: kernel wall time, per-workgroup setup / prologue / K loop / epilogue (100 MHz wall clock), and per CU the gap between
the end of one workgroup and the start of the next."""
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
          ('convT 256->128 in 80x208', lambda: hnn.ConvTranspose2d(256, 128, 6, 2, 2), (20, 256, 80, 208)),
          ('convT 512->256 in 40x104', lambda: hnn.ConvTranspose2d(512, 256, 6, 2, 2), (20, 512, 40, 104))]
LAYERS.append(('conv 64->64 3x3 in 80x208 (four-wave 64x128 tile)', lambda: hnn.Conv2d(64, 64, 3, 1, 1), (20, 64, 80, 208)))
LAYERS.append(('conv 128->128 3x3 in 40x104', lambda: hnn.Conv2d(128, 128, 3, 1, 1), (20, 128, 40, 104)))
NK = {LAYERS[0][0]: 36, LAYERS[1][0]: 72, LAYERS[2][0]: 144, LAYERS[3][0]: 18, LAYERS[4][0]: 36}
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if sys.argv[1] in l[0]]
for name, make, shape in LAYERS:
    torch.manual_seed(0)
    with torch.device(dev):
        m = make()
    x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        assert L.muvo_debug_bf3_stamps_reset() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = m(x); b.record()
        torch.cuda.synchronize()
    buf = np.zeros(8 * 16384, dtype=np.uint64)
    assert L.muvo_debug_bf3_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(16384, 8)
    st = st[st[:, 0] > 0]
    t = st[:, [0, 1, 2, 3, 4, 6]].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    hw = st[:, 5]
    cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 13) & 0x7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)   # xcc, se, sh, cu
    print(f'== {name}: {len(st)} workgroups (incl. split pass launches before: event time {a.elapsed_time(b):.3f} ms), '
          f'kernel span {us[:, 4].max():.1f} us, {len(np.unique(cu))} distinct CUs')
    for lbl, d in (('setup', us[:, 1] - us[:, 0]), ('prologue', us[:, 2] - us[:, 1]), ('K loop', us[:, 3] - us[:, 2]),
                   ('epilogue issue', us[:, 4] - us[:, 3]), ('whole workgroup', us[:, 4] - us[:, 0])):
        print(f'   {lbl:16s} mean {d.mean():8.2f}  p10 {np.percentile(d, 10):8.2f}  median {np.median(d):8.2f}  p90 {np.percentile(d, 90):8.2f} us')
    if (t[:, 5] > 0).any():
        d = us[:, 5] - us[:, 4]
        print(f'   {"store drain":16s} mean {d.mean():8.2f}  p10 {np.percentile(d, 10):8.2f}  median {np.median(d):8.2f}  p90 {np.percentile(d, 90):8.2f} us')
    loop_clk = st[:, 7].astype(np.float64)
    loop_us = us[:, 3] - us[:, 2]
    mhz = loop_clk / np.maximum(loop_us, 1e-3)
    print(f'   s_memtime ticks per us over the K loop: median {np.median(mhz):.0f} (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f});  '
          f'ticks per K step {np.median(loop_clk) / NK[name]:.0f}')
    gaps, per_cu = [], []
    for c in np.unique(cu):
        sel = us[cu == c]
        sel = sel[np.argsort(sel[:, 0])]
        per_cu.append(len(sel))
        gaps += list(sel[1:, 0] - sel[:-1, 4])
    gaps = np.array(gaps)
    print(f'   workgroups per CU {min(per_cu)}..{max(per_cu)};  gap end -> next start on the same CU: mean {gaps.mean():.2f} median '
          f'{np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us')
    # residency: time-average of the number of workgroups between their first and last stamp on one CU, and the hand-over
    # delay of a slot: start of the (k + R)-th workgroup of a CU minus the end stamp of its k-th, R = residents the LDS allows
    R = 1 if 'four-wave' not in name else 3
    res, hand = [], []
    for c in np.unique(cu):
        sel = us[cu == c]
        sel = sel[np.argsort(sel[:, 0])]
        res.append((sel[:, 4] - sel[:, 0]).sum() / (sel[:, 4].max() - sel[:, 0].min()))
        ends = np.sort(sel[:, 4])
        if len(sel) > R:
            hand += list(sel[R:, 0] - ends[:-R])
    hand = np.array(hand)
    print(f'   resident workgroups per CU (time average between a CU\'s first start and last end): mean {np.mean(res):.2f};  slot hand-over '
          f'(start of workgroup k+{R} - k-th end on the CU): median {np.median(hand):.2f} p10 {np.percentile(hand, 10):.2f} p90 {np.percentile(hand, 90):.2f} us')
    first = np.sort(us[:, 0])[:256]
    print(f'   start of the first 256 workgroups: {first.min():.1f} .. {first.max():.1f} us;  last end {us[:, 4].max():.1f} us')
