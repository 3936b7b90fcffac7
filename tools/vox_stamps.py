"""Shader-clock ticks per plane that wave 0 of a vox_bf3_wgrad workgroup spends staging (split + LDS writes), at the barriers,
issuing the next loads and in the MFMA phase.  tools/ab_local.sh vst "-DMUVO_VOX_STAMPS=1" conv_vox.hip, then on the GPU box
MUVO_HIP_LIB=muvo_amd/build_ab/vst/libmuvo_hip.so python tools/vox_stamps.py"""
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
L.muvo_debug_vox_stamps.argtypes = [C.c_void_p, C.c_int]
for name, cin, cout, shape in (('16->8 192x192x64', 16, 8, (20, 16, 192, 192, 64)), ('8->8 192x192x64', 8, 8, (20, 8, 192, 192, 64)),
                               ('32->16 96x96x32', 32, 16, (20, 32, 96, 96, 32))):
    torch.manual_seed(0)
    with torch.device(dev):
        m = hnn.Conv3d(cin, cout, 3, 1, 1)
    x = torch.randn(*shape, device=dev, requires_grad=True)
    y = m(x)
    g = torch.randn_like(y)
    for _ in range(2):
        y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 1024, dtype=np.uint64)
    assert L.muvo_debug_vox_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(1024, 8).astype(np.float64)
    st = st[st[:, 5] > 0]
    per = st[:, :5] / st[:, 5:6]
    print(f'== {name}: {len(st)} workgroups, {st[0, 5]:.0f} planes each; ticks per plane (median over workgroups):')
    for lbl, col in zip(('stage (split + LDS writes)', 'barrier after staging', 'issue next loads', 'MFMA phase (LDS reads, alignbyte, MFMA)', 'barrier after MFMA'), per.T):
        print(f'   {lbl:42s} {np.median(col):9.0f}')
    print(f'   {"total":42s} {np.median(per.sum(axis=1)):9.0f}')
