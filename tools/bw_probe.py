import torch, time
d=torch.device('cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
x=torch.empty(3*1024**3//4, device=d); y=torch.empty_like(x)
ms=t(lambda: x.fill_(1.0)); print('fill 3 GiB      %.3f ms  %.2f TB/s'%(ms, 3*1.0737/ms))
ms=t(lambda: y.copy_(x)); print('copy 3+3 GiB    %.3f ms  %.2f TB/s'%(ms, 6*1.0737/ms))
ms=t(lambda: x.sum()); print('sum 3 GiB       %.3f ms  %.2f TB/s'%(ms, 3*1.0737/ms))
ms=t(lambda: torch.add(x, 1.0, out=y)); print('add 3+3 GiB     %.3f ms  %.2f TB/s'%(ms, 6*1.0737/ms))
ms=t(lambda: torch.cuda.memset if False else x.zero_()); print('zero 3 GiB      %.3f ms  %.2f TB/s'%(ms, 3*1.0737/ms))
