"""Which part of the 1x1 head kernel goes wrong next to a big-LDS convolution workgroup of another stream?  Builds
tools/dev/coresidency_victim.hip on the box and runs its victim kernel (weights from LDS or from global memory, with an in-kernel
cross-check) on the current stream while stream A loops a bf16x3 Conv3d (vox_bf3_kernel, 127 KB of LDS per workgroup).
    python tools/dev/coresidency_repro2.py [aggressor=vox|conv|none] [iters]"""
import ctypes as C, os, subprocess, sys, torch
sys.path.insert(0, '/root/repo')
from muvo_amd import nn as hnn, ops

here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/coresidency_victim.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so,
                       os.path.join(here, 'coresidency_victim.hip')] + os.environ.get('VICTIM_FLAGS', '').split())
vl = C.CDLL(so)
vl.victim_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
vflags = os.environ.get('VICTIM_FLAGS', '')
aggressor = sys.argv[1] if len(sys.argv) > 1 else 'vox'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
torch.manual_seed(0)
ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
with torch.device(dev):
    conv = hnn.Conv2d(128, 128, 3, 1, 1, bias=True)
    vox = hnn.Conv3d(32, 32, 3, 1, 1, bias=True)
xa = torch.randn(20, 128, 80, 400, device=dev)
xv3 = torch.randn(2, 32, 96, 96, 32, device=dev)
N, Cin, H, W = 20, 64, 160, 800
xv = torch.randn(N, Cin, H, W, device=dev)
w = torch.randn(3, Cin, device=dev) * 0.1
S4 = H * W // 4


def run_aggressor():
    if aggressor == 'conv':
        conv(xa, act=1)
    elif aggressor == 'vox':
        vox(xv3, act=2, slope=0.2)


def victim(mode, stats, lds_extra=0):
    out = torch.empty(N, 3, H, W, device=dev)
    rc = vl.victim_launch(xv.data_ptr(), w.data_ptr(), out.data_ptr(), N, Cin, S4, mode, stats.data_ptr(), lds_extra,
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


with torch.no_grad():
    side = torch.cuda.Stream(device=dev)
    for mode, extra, label in ((0, 0, 'weights from global memory'), (1, 0, 'weights from LDS'),
                               (3, 0, 'weights from LDS + in-kernel cross-check'), (1, 40960, 'weights from LDS, 40 KB reserved'),
                               (0, 40960 * 0 + 1024, 'weights from global memory, LDS allocated but unused')):
        stats = torch.zeros(4, dtype=torch.int32, device=dev)
        run_aggressor(); ref = victim(mode, stats, extra).clone(); torch.cuda.synchronize()
        stats.zero_()
        bad = 0
        for it in range(iters):
            with torch.cuda.stream(side):
                for _ in range(3):
                    run_aggressor()
            outs = [victim(mode, stats, extra) for _ in range(4)]
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print(f'aggressor={aggressor} [{vflags}] {label}: {bad} of {4 * iters} launches differ; in-kernel: '
              f'{int(stats[0])} lanes differ from the global-weight evaluation, {int(stats[1])} LDS words wrong at the end, '
              f'{int(stats[2])} LDS words wrong right after the write')
