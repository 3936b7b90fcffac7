"""BEV lifting operator at base_1d sizes (20 frames, 384 channels, 40x104 feature map, 37 depth bins, top-10 mask):
HIP forward / backward time against the HBM roofline of its algorithmic bytes, and the CPU oracle timed beside it.
    python tools/bev_bench.py [--frames 20] [--iters 20]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
from muvo_amd import bev  # noqa: E402
from muvo_amd.data.frustum_inputs import frustum_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=20)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cpu-frames', type=int, default=2)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    B, C, H, W = args.frames, 384, 40, 104
    c = frustum_case(B=B, C=C, H=H, W=W, key='frustum_bench')
    a = dict(size=c['size'], scale=c['scale'], offsetx=c['offsetx'], dbound=c['dbound'], downsample=c['downsample'])
    pool = bev.FrustumPooling(**a).to(dev)
    feat, depth = c['feat'].to(dev).requires_grad_(True), c['depth'].to(dev).requires_grad_(True)
    intr, ext, mask, gout = c['intrinsics'].to(dev), c['extrinsics'].to(dev), c['mask'].to(dev), c['gout'].to(dev)
    D, ncell = pool.D, pool.nx_constant[0] * pool.nx_constant[1]
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for it in range(args.iters + 3):
        feat.grad = depth.grad = None
        e0.record()
        out = pool.lift(feat, depth, intr, ext, mask)
        e1.record()
        out.backward(gout)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
    tf, tb = tf / args.iters, tb / args.iters
    lifted = int(mask.sum())
    # algorithmic HBM bytes: feature map + lifted depth values + cell indices + mask in, BEV map out (backward: the same
    # inputs + the BEV gradient in, both gradients out)
    fwd_bytes = 4 * B * C * H * W + B * D * H * W * (4 + 4 + 1) + 4 * B * C * ncell
    bwd_bytes = 4 * B * C * H * W + B * D * H * W * (4 + 4 + 1) + 4 * B * C * ncell + 4 * B * C * H * W + 4 * B * D * H * W
    print(f'frames {B}: lifted points {lifted} ({lifted / B:.0f} per frame), BEV cells {ncell}, channels {C}')
    print(f'forward  {tf:.3f} ms  {fwd_bytes / tf / 1e6:.1f} GB/s algorithmic ({fwd_bytes / 1e6:.1f} MB; {100 * fwd_bytes / tf / 1e6 / 8000:.1f} % of 8 TB/s), '
          f'{lifted * C / tf / 1e6:.1f} G atomic adds/s')
    print(f'backward {tb:.3f} ms  {bwd_bytes / tb / 1e6:.1f} GB/s algorithmic ({bwd_bytes / 1e6:.1f} MB; {100 * bwd_bytes / tb / 1e6 / 8000:.1f} % of 8 TB/s)')
    from oracle import muvo_ref as R
    n = args.cpu_frames
    t0 = time.perf_counter()
    R.frustum_pool(c['feat'][:n], c['depth'][:n], c['mask'][:n], c['intrinsics'][:n], c['extrinsics'][:n], **a)
    t1 = time.perf_counter()
    print(f'cpu oracle forward: {1e3 * (t1 - t0) / n:.1f} ms per frame on {torch.get_num_threads()} threads '
          f'({(t1 - t0) / n / (tf * 1e-3 / B):.0f}x the HIP forward per frame)')


if __name__ == '__main__':
    main()
