import sys, torch
sys.path.insert(0, '/root/repo')
from muvo_amd import ops
dev = torch.device('cuda:0')
for rows, out_f, in_f in ((20, 512, 33280), (20, 512, 8192), (20, 128, 1536), (2, 3072, 1024), (20, 16, 1536)):
    x = torch.randn(rows, in_f, device=dev); w = torch.randn(out_f, in_f, device=dev) * 0.01; b = torch.randn(out_f, device=dev)
    y = ops.linear(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.linear(x, w, b)
    e1.record(); torch.cuda.synchronize()
    print(rows, out_f, in_f, f'{e0.elapsed_time(e1) * 20:.1f} us/call  rel err {err:.1e}  ptr&15 {x.data_ptr() & 15} {w.data_ptr() & 15}')
