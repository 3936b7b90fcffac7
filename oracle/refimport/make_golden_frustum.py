"""Golden fixture of the BEV lifting operator: runs the REAL reference's FrustumPooling
(muvo/models/frustum_pooling.py:67-209) forward and backward, in training mode (QuickCumsum) and with the sparse top-k
depth mask of mile.py:509-518, on a deterministic small case, and writes tests/golden/frustum_pool.{json,npz}.

The reference sums the features of a BEV cell with a float32 cumsum over ALL lifted points followed by differences
(frustum_pooling.py:23-53), so its own output carries rounding noise that grows with the prefix sum; the fixture therefore
also stores the same module evaluated in float64 (`out64`), which is the value the rounding-free algorithm produces.

Usage: python oracle/refimport/make_golden_frustum.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.frustum_inputs import frustum_case  # noqa: E402


def run(FP, c, dtype):
    fp = FP(size=c['size'], scale=c['scale'], offsetx=c['offsetx'], dbound=c['dbound'], downsample=c['downsample']).to(dtype)
    fp.train()
    feat = c['feat'].detach().clone().to(dtype).requires_grad_(True)
    depth = c['depth'].detach().clone().to(dtype).requires_grad_(True)
    x = (depth.unsqueeze(1) * feat.unsqueeze(2))                  # mile.py:519 outer product (B, C, D, H, W)
    x = x.unsqueeze(1).permute(0, 1, 3, 4, 5, 2)                  # (B, 1, D, H, W, C)
    out = fp(x, c['intrinsics'].to(dtype).unsqueeze(1), c['extrinsics'].to(dtype).unsqueeze(1), c['mask'])
    (out * c['gout'].to(dtype)).sum().backward()
    return out.detach(), feat.grad, depth.grad, fp


def main():
    G.import_reference()
    from muvo.models.frustum_pooling import FrustumPooling
    c = frustum_case()
    out32, dfeat32, ddepth32, fp = run(FrustumPooling, c, torch.float32)
    out64, dfeat64, ddepth64, _ = run(FrustumPooling, c, torch.float64)
    print('out', tuple(out32.shape), 'nonzero cells', int((out64 != 0).any(1).sum()), 'of', out64.shape[0] * out64.shape[2] * out64.shape[3])
    print('reference fp32 vs its fp64 run: out %.3e  dfeat %.3e  ddepth %.3e (max abs / max)' % (
        float((out32 - out64).abs().max() / out64.abs().max()), float((dfeat32 - dfeat64).abs().max() / dfeat64.abs().max()),
        float((ddepth32 - ddepth64).abs().max() / ddepth64.abs().max())))
    dm = fp.get_depth_map(c['depth'])
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'frustum_pool.npz'), out32=out32.numpy(), out64=out64.numpy(),
                        dfeat64=dfeat64.numpy(), ddepth64=ddepth64.numpy(), dfeat32=dfeat32.numpy(), ddepth32=ddepth32.numpy(),
                        depth_map=dm.numpy())
    with open(os.path.join(REPO, 'tests', 'golden', 'frustum_pool.json'), 'w') as f:
        json.dump(dict(nx=fp.nx_constant, D=fp.D, out_shape=list(out32.shape),
                       ref32_vs_ref64=dict(out=float((out32 - out64).abs().max()), out_max=float(out64.abs().max()))), f)
    print('wrote tests/golden/frustum_pool.*')


if __name__ == '__main__':
    main()
