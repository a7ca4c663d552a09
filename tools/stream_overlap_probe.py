"""Do two HIP streams of one process overlap on this box?  A chain of tiny launch-bound kernels on one stream
and one bandwidth-bound fill on another, issued back to back, against the same work on a single stream."""
import time

import torch

big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")      # 1 GiB fill: ~160 us
small = [torch.zeros(1024, device="cuda") for _ in range(40)]
side = torch.cuda.Stream()


def chain():
    for t in small:
        t.add_(1.0)


def serial():
    chain()
    big.fill_(1.0)


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        big.fill_(1.0)
    chain()
    main.wait_stream(side)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("chain alone   %.1f us" % t(chain))
print("fill alone    %.1f us" % t(lambda: big.fill_(1.0)))
print("one stream    %.1f us" % t(serial))
print("two streams   %.1f us" % t(forked))
g = torch.cuda.CUDAGraph()
forked(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    forked()
print("two streams, graph replay %.1f us" % t(g.replay))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    serial()
print("one stream, graph replay  %.1f us" % t(g2.replay))
