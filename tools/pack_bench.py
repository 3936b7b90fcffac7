"""Time of the batched weight re-pack (pack_table_kernel) per layer type: one layer in the table at a time.
   python tools/pack_bench.py   (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muvo_amd import nn as hnn, ops

dev = torch.device('cuda', 0)
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
LAYERS = [('convT 512->512 k6', lambda: hnn.ConvTranspose2d(512, 512, 6, 2, 2), (2, 512, 20, 52)),
          ('convT 512->512 k5', lambda: hnn.ConvTranspose2d(512, 512, 5, 2, 2, 1), (2, 512, 10, 26)),
          ('convT 512->256 k6', lambda: hnn.ConvTranspose2d(512, 256, 6, 2, 2), (2, 512, 40, 104)),
          ('convT 128->64 k6', lambda: hnn.ConvTranspose2d(128, 64, 6, 2, 2), (2, 128, 80, 104)),
          ('conv 512->512 3x3', lambda: hnn.Conv2d(512, 512, 3, 1, 1), (4, 512, 20, 52)),
          ('conv 256->512 3x3 s2', lambda: hnn.Conv2d(256, 512, 3, 2, 1), (4, 256, 40, 104)),
          ('conv 64->64 3x3', lambda: hnn.Conv2d(64, 64, 3, 1, 1), (4, 64, 80, 208)),
          ('conv3d 64->64', lambda: hnn.Conv3d(64, 64, 3, 1, 1), (2, 64, 24, 24, 8)),
          ('linear 384->1536', lambda: hnn.Linear(384, 1536), (6480, 384))]
for name, make, shape in LAYERS:
    ops._PACKS.__init__()
    torch.manual_seed(0)
    with torch.device(dev):
        m = make()
    x = torch.randn(*shape, device=dev, requires_grad=True)
    y = m(x)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    for _ in range(3):
        ops.repack_all()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.repack_all()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nw = m.weight.numel()
    print(f'{name:24s} {nw / 1e6:6.2f} M weights  {us:8.1f} us  {nw * 16 / us / 1e6:6.2f} TB/s (both packed forms: 8 B read + 8 B written per weight)  items {ops._PACKS.n} blocks {ops._PACKS.nblk}')
