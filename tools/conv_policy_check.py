"""Every conv layer shape of a layer table (bench.py --layer-table) under the library's family-choice policy against the
exact-fp32 mode of the same kernels: relative error of forward (with ELU: smooth, so that rounding differences cannot flip a mask), data gradient, weight and bias gradient.
    python tools/conv_policy_check.py profiles/r01p_bf16x3_conv_layers.txt [batch]"""
import os, re, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muvo_amd import nn as hnn, ops  # noqa: E402

PAT = re.compile(r'(convT|conv)(\d)d (\d+)->(\d+) k\((\d+), (\d+), (\d+)\) s\((\d+), (\d+), (\d+)\) n(\d+) in\((\d+), (\d+), (\d+)\)')


def rel(a, b):
    return ((a.double() - b.double()).pow(2).sum().sqrt() / b.double().pow(2).sum().sqrt().clamp_min(1e-30)).item()


def main():
    dev = torch.device('cuda:0')
    shapes = sorted({m.groups() for m in map(PAT.search, open(sys.argv[1])) if m})
    for c_in, c_out, isz in ((384, 512, (48, 48)), (512, 384, (24, 24)), (384, 512, (24, 24)), (64, 128, (40, 104)), (256, 64, (20, 52))):
        shapes.append(('conv', '2', str(c_in), str(c_out), '1', '5', '5', '1', '2', '2', '20', '1', str(isz[0]), str(isz[1])))
    for c_in, c_out, isz in ((512, 64, (10, 26)), (256, 64, (20, 52)), (128, 64, (40, 104)), (64, 64, (80, 208))):
        shapes.append(('conv', '2', str(c_in), str(c_out), '1', '1', '1', '1', '1', '1', '20', '1', str(isz[0]), str(isz[1])))
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    worst = 0.0
    for tr, nd, cin, cout, k0, k1, k2, s0, s1, s2, n, i0, i1, i2 in shapes:
        nd, cin, cout = int(nd), int(cin), int(cout)
        k, s, isz = (int(k0), int(k1), int(k2)), (int(s0), int(s1), int(s2)), (int(i0), int(i1), int(i2))
        if nd == 2:
            k, s, isz = k[1:], s[1:], isz[1:]
        pad = tuple((kk - 1) // 2 if tr == 'conv' else (kk - ss) // 2 for kk, ss in zip(k, s))
        out = {}
        for mode in (ops.CONV_F32, ops.CONV_BF16X3):
            ops.set_conv_mode(mode, min_gflop=-1.0)
            torch.manual_seed(1)
            with torch.device(dev):
                cls = {('conv', 2): hnn.Conv2d, ('conv', 3): hnn.Conv3d, ('convT', 2): hnn.ConvTranspose2d}[(tr, nd)]
                m = cls(cin, cout, k, s, pad)
                x = torch.randn(nb, cin, *isz).requires_grad_(True)
            y = m(x, act=ops.ACT_ELU)
            torch.manual_seed(2)
            g = torch.randn_like(y)
            m.weight.grad, m.bias.grad = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
            y.backward(g)
            key = (nb, (1,) * (3 - nd) + tuple(isz), ops._plan_epoch[0])
            out[mode] = (y.detach(), x.grad, m.weight.grad.clone(), m.bias.grad.clone(), m.geom.family.get(key))
            del m, x, y, g
        a, b = out[ops.CONV_F32], out[ops.CONV_BF16X3]
        errs = [rel(b[i], a[i]) for i in range(4)]
        worst = max(worst, *errs)
        flag = '  <-- ' if max(errs) > 2e-4 else ''
        print(f'{tr}{nd}d {cin}->{cout} k{k} s{s} in{isz}: family {b[4]}  y {errs[0]:.1e} dx {errs[1]:.1e} dw {errs[2]:.1e} '
              f'db {errs[3]:.1e}{flag}', flush=True)
    print('worst', worst)


if __name__ == '__main__':
    main()
