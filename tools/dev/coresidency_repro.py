"""Does a small kernel on one stream compute wrong values while eight-wave bf16x3 convolution tiles run on another stream?
(profiles/r03j_lidar_decoder_stream.txt.)  Stream A loops a 3x3 convolution that runs on the 256x128 eight-wave tile (150 KB of
LDS per workgroup); the current stream runs a victim kernel (the 1x1 head, an activation, ...) over and over and compares every
result bit for bit with the result it gave on an idle GPU.
    python tools/dev/coresidency_repro.py [victim=head|head_nobias|head_direct|act] [aggressor=vox|conv|gemm|gemm32|ew|none] [iters]
The shipped library (built without packed fp32 instructions, muvo_amd/build.py) gives 0 differing launches.  To see the failure,
build a library WITH them and point MUVO_HIP_LIB at it:
    MUVO_HIPCC_EXTRA="-Xclang -target-feature -Xclang +packed-fp32-ops" python -c "from muvo_amd.build import build; build(force=True)"
(every run is listed in profiles/r03j_lidar_decoder_stream.txt)."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from muvo_amd import nn as hnn, ops

victim = sys.argv[1] if len(sys.argv) > 1 else 'head'
aggressor = sys.argv[2] if len(sys.argv) > 2 else 'conv'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device('cuda:0')
torch.manual_seed(0)
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
with torch.device(dev):
    conv = hnn.Conv2d(128, 128, 3, 1, 1, bias=True)
    vox = hnn.Conv3d(32, 32, 3, 1, 1, bias=True)
    head = hnn.Conv2d(64, 3, 1, 1, 0, bias=(victim != 'head_nobias'))
xa = torch.randn(20, 128, 80, 400, device=dev)
xv3 = torch.randn(2, 32, 96, 96, 32, device=dev)
xv = torch.randn(20, 64, 160, 800, device=dev)
ga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); gb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
ga32 = torch.randn(4096, 4096, device=dev); gb32 = torch.randn(4096, 4096, device=dev)


import ctypes as C
_geom = ops.ConvGeom(2, 0, 64, 3, 1)
_desc = _geom.plan(20, (1, 160, 800))[0]
_ybuf = [torch.empty(20, 3, 160, 800, device=dev) for _ in range(6)]
_yi = [0]
_wflat = head.weight.detach().reshape(3, 64).contiguous()


def run_victim():
    if victim == 'head_direct':      # the C entry point itself: no autograd Function, no allocation, no weight-copy check
        y = _ybuf[_yi[0] % 6]; _yi[0] += 1
        ops._ck(ops.lib().muvo_conv_forward(C.byref(_desc), ops._f(xv), ops._f(_wflat), ops._f(head.bias.detach()), ops._f(y), 0,
                                            ops._fl(0.0), None, ops._st()))
        return y
    if victim == 'act':
        return ops.activation(xv, 2, 0.2)
    return head(xv)


def run_aggressor():
    if aggressor == 'conv':
        conv(xa, act=1)
    elif aggressor == 'vox':
        vox(xv3, act=2, slope=0.2)
    elif aggressor == 'gemm':       # rocBLAS / hipBLASLt bf16 MFMA kernels: nothing of this repository
        torch.matmul(ga, gb)
    elif aggressor == 'gemm32':
        torch.matmul(ga32, gb32)
    elif aggressor == 'ew':
        torch.add(xa, 1.0)


with torch.no_grad():
    run_aggressor(); ref = run_victim().clone(); torch.cuda.synchronize()
    again = run_victim().clone(); torch.cuda.synchronize()
    assert torch.equal(ref, again), 'victim is not deterministic on an idle GPU'
    side = torch.cuda.Stream(device=dev)
    bad = 0
    worst = 0.0
    for it in range(iters):
        with torch.cuda.stream(side):
            for _ in range(3):
                run_aggressor()
        outs = [run_victim() for _ in range(4)]
        torch.cuda.synchronize()
        if victim == 'head_direct':
            outs = [o.clone() for o in outs]
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                d = (o - ref).abs()
                worst = max(worst, float(d.max()))
                if bad <= 3:
                    idx = torch.nonzero(d.flatten() > 0).flatten()
                    print(f'  iter {it}: {idx.numel()} values differ, max {float(d.max()):.3e}, first flat indices {idx[:8].tolist()}')
                    if o.dim() == 4 and o.shape[1] <= 4:
                        S = o.shape[2] * o.shape[3]
                        co = (idx // S) % o.shape[1]
                        comp = idx % 4
                        lane16 = ((idx % S) // 4) % 64 // 16
                        blk = (idx % S) // 4 // 256
                        print('    by channel', torch.bincount(co, minlength=o.shape[1]).tolist(), 'by float4 component',
                              torch.bincount(comp, minlength=4).tolist(), 'by 16-lane group', torch.bincount(lane16, minlength=4).tolist(),
                              'workgroups touched', int(torch.unique(blk + 4096 * (idx // (S * o.shape[1]))).numel()))
                        W_ = head.weight.view(o.shape[1], -1); X_ = xv.view(xv.shape[0], xv.shape[1], -1)
                        hist = {}
                        for j in idx[torch.randperm(idx.numel(), device=idx.device)[:400]].tolist():
                            dd = (o.flatten()[j] - ref.flatten()[j]).item()
                            nn_, cc, pp = j // (S * o.shape[1]), (j // S) % o.shape[1], j % S
                            terms = W_[cc] * X_[nn_, :, pp]
                            e = (terms + dd).abs()              # got = want - term  ->  term + diff = 0
                            ci = int(e.argmin())
                            key = (ci % 4 if float(e[ci]) < 1e-5 else 'no single term', cc)
                            hist[key] = hist.get(key, 0) + 1
                        print('    missing term by (ci % 4, channel):', sorted(hist.items(), key=lambda kv: str(kv[0])))
                        j = int(idx[0]); dd = (o.flatten()[j] - ref.flatten()[j]).item()
                        nn_, cc, pp = j // (S * o.shape[1]), (j // S) % o.shape[1], j % S
                        terms = head.weight.view(o.shape[1], -1)[cc] * xv.view(xv.shape[0], xv.shape[1], -1)[nn_, :, pp]
                        print(f'    first: got {o.flatten()[j].item():.6f} want {ref.flatten()[j].item():.6f} diff {dd:.6f}; closest single term '
                              f'{terms[(terms.abs() - abs(dd)).abs().argmin()].item():.6f}, bias {head.bias[cc].item() if head.bias is not None else 0:.6f}')
    print(f'victim={victim} aggressor={aggressor} PW_LDS_RESERVE={os.environ.get("MUVO_PW_LDS_RESERVE", "default")}: '
          f'{bad} of {4 * iters} victim launches differ (max {worst:.3e})')
