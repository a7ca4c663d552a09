import torch, time
x = torch.empty(1<<28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
ms = t(lambda: x.fill_(1.0)); print("fill 1 GiB  %.1f us  %.2f TB/s write" % (ms*1e3, x.numel()*4/ms/1e9))
ms = t(lambda: y.copy_(x)); print("copy 1 GiB  %.1f us  %.2f TB/s r+w" % (ms*1e3, 2*x.numel()*4/ms/1e9))
xs = x[:96*1024*1024]; ys = y[:96*1024*1024]
ms = t(lambda: xs.fill_(1.0)); print("fill 384 MiB %.1f us  %.2f TB/s write" % (ms*1e3, xs.numel()*4/ms/1e9))
ms = t(lambda: ys.copy_(xs)); print("copy 384 MiB %.1f us  %.2f TB/s r+w" % (ms*1e3, 2*xs.numel()*4/ms/1e9))
ms = t(lambda: torch.sum(xs)); print("read 384 MiB %.1f us  %.2f TB/s read" % (ms*1e3, xs.numel()*4/ms/1e9))
