"""One conv layer forward + backward in a loop (for rocprofv3 --kernel-trace --stats): which kernels does a shape launch?
    python tools/one_conv.py conv2d 128 256 3 2 1 20 8 8      (kind cin cout k stride pad batch H W [D])"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd import nn as hnn, ops
kind, cin, cout, k, s, p, n = sys.argv[1], *map(int, sys.argv[2:8])
sz = tuple(map(int, sys.argv[8:]))
dev = torch.device('cuda:0')
with torch.device(dev):
    m = {'conv2d': hnn.Conv2d, 'conv3d': hnn.Conv3d, 'convT2d': hnn.ConvTranspose2d}[kind](cin, cout, k, s, p)
    x = torch.randn(n, cin, *sz).requires_grad_(True)
m.weight.grad, m.bias.grad = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
for it in range(12):
    if it == 2:
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    y = m(x, act=ops.ACT_RELU)
    y.backward(torch.ones_like(y))
e1.record(); torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) * 100:.1f} us per fwd+bwd')
