"""Accuracy of the two conv arithmetics (exact fp32 MFMA, bf16x3 split products) against an fp64 CPU reference:
relative RMS and max error of forward, data-gradient and weight-gradient of one decoder-like layer."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd import nn as hnn  # noqa: E402
from muvo_amd import ops  # noqa: E402


def stats(name, got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    e = (got - ref)
    print(f'    {name:6s} rms_err/rms_ref {e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt():.2e}   '
          f'max_err/max_ref {e.abs().max() / ref.abs().max():.2e}')


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cases = [('convT 128->64 k6s2 20x26', True, 128, 64, 6, 2, 2, (20, 26)),
             ('conv 64->64 k3 24x40', False, 64, 64, 3, 1, 1, (24, 40)),
             ('convT 512->256 k6s2 5x13', True, 512, 256, 6, 2, 2, (5, 13))]
    for title, tr, cin, cout, k, s, p, sz in cases:
        print(title)
        x = torch.randn(2, cin, *sz)
        for mode, mname in ((ops.CONV_F32, 'f32'), (ops.CONV_BF16X3, 'bf16x3')):
            ops.set_conv_mode(mode)
            torch.manual_seed(1)
            with torch.device(dev):
                m = hnn.ConvTranspose2d(cin, cout, k, s, p) if tr else hnn.Conv2d(cin, cout, k, s, p)
            w64 = m.weight.detach().cpu().double().requires_grad_(True)
            b64 = m.bias.detach().cpu().double()
            x64 = x.double().requires_grad_(True)
            y64 = F.conv_transpose2d(x64, w64, b64, s, p) if tr else F.conv2d(x64, w64, b64, s, p)
            g = torch.randn_like(y64)
            y64.backward(g)
            xg = x.to(dev).requires_grad_(True)
            y = m(xg)
            m.weight.grad = torch.zeros_like(m.weight)
            m.bias.grad = torch.zeros_like(m.bias)
            y.backward(g.float().to(dev))
            print(f'  {mname}')
            stats('fwd', y, y64)
            stats('dgrad', xg.grad, x64.grad)
            stats('wgrad', m.weight.grad, w64.grad)
        # CPU fp32 for scale
        w32 = w64.detach().float().requires_grad_(True)
        x32 = x.clone().requires_grad_(True)
        y32 = F.conv_transpose2d(x32, w32, b64.float(), s, p) if tr else F.conv2d(x32, w32, b64.float(), s, p)
        y32.backward(g.float())
        print('  torch CPU fp32')
        stats('fwd', y32, y64)
        stats('dgrad', x32.grad, x64.grad)
        stats('wgrad', w32.grad, w64.grad)


if __name__ == '__main__':
    main()
