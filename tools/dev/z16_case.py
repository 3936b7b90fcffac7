"""dev: the Z = 16 Conv3d cases of test_conv_family, errors per gradient, under the current MUVO_VOX_WGRAD_PS setting"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muvo_amd import nn as hnn, ops
dev = torch.device('cuda:0')
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
for cin, cout, sz, n in ((32, 32, (4, 31, 16), 3), (32, 32, (6, 8, 16), 3), (32, 32, (3, 35, 16), 3), (64, 32, (5, 19, 16), 3), (16, 8, (3, 16, 16), 3), (32, 32, (3, 7, 32), 3)):
    torch.manual_seed(0)
    with torch.device(dev):
        m = hnn.Conv3d(cin, cout, 3, 1, 1, bias=True)
    x = torch.randn(n, cin, *sz)
    xg = x.to(dev).requires_grad_(True)
    y = m(xg, act=2, slope=0.2)
    w = m.weight.detach().cpu().requires_grad_(True); b = m.bias.detach().cpu().requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    yr = F.leaky_relu(F.conv3d(xc, w, b, 1, 1), 0.2)
    g = torch.randn_like(yr)
    yr.backward(g)
    m.weight.grad = torch.zeros_like(m.weight); m.bias.grad = torch.zeros_like(m.bias)
    y.backward(g.to(dev))
    fam = [ops.lib().muvo_conv_kernel_family(__import__('ctypes').byref(m.geom.plan(n, sz)[0]), op) for op in (0, 1, 2)] if hasattr(m, 'geom') else None
    e = lambda a, r: f'{(a.detach().cpu() - r).abs().max().item():.2e}/{r.abs().max().item():.2e}'
    print(cin, cout, sz, n, 'family', fam, 'fwd', e(y, yr), 'dgrad', e(xg.grad, xc.grad), 'wgrad', e(m.weight.grad, w.grad), 'dbias', e(m.bias.grad, b.grad), flush=True)
