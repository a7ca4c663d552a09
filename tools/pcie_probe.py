import torch, time
for mb in (2, 36, 164):
    n = mb * 1024 * 1024 // 4
    h = torch.empty(n, dtype=torch.float32).pin_memory()
    d = torch.empty(n, dtype=torch.float32, device="cuda")
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%s %4d MB  %.3f ms  %.1f GB/s" % (name, mb, dt * 1e3, mb / 1024 / dt))
