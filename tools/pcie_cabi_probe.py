#!/usr/bin/env python3
"""Host <-> device copy rate through libclapgpu's own C-ABI helpers (clapgpu_host_malloc = hipHostMalloc,
clapgpu_memcpy_* = hipMemcpyAsync on the null stream), the calls libclapgpu_scene makes; compare tools/pcie_probe.py
(torch pinned tensors)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clap_amd import _lib
L = _lib.lib()
_lib.check(L.clapgpu_init(0), "init")
for mb in (2, 36, 172):
    n = mb << 20
    h, d = C.c_void_p(), C.c_void_p()
    _lib.check(L.clapgpu_host_malloc(C.byref(h), n), "host_malloc")
    _lib.check(L.clapgpu_malloc(C.byref(d), n), "malloc")
    C.memset(h, 1, n)
    for name, fn, a, b in (("h2d", L.clapgpu_memcpy_h2d, d, h), ("d2h", L.clapgpu_memcpy_d2h, h, d)):
        for _ in range(2):
            fn(a, b, n, None); L.clapgpu_stream_sync(None)
        t0 = time.perf_counter()
        for _ in range(5):
            fn(a, b, n, None); L.clapgpu_stream_sync(None)
        dt = (time.perf_counter() - t0) / 5
        print(f"{mb:4d} MB {name}: {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.1f} GB/s")
    t0 = time.perf_counter(); C.memset(h, 2, n); dt = time.perf_counter() - t0
    print(f"{mb:4d} MB host memset of the pinned buffer: {dt*1e3:.3f} ms {n/dt/1e9:.1f} GB/s")
    L.clapgpu_free(d); L.clapgpu_host_free(h)

# CPU-side read rate of the page-locked buffer (what a binding's scatter-back pass sees) vs ordinary memory
import numpy as np
n = 172 << 20
h = C.c_void_p()
_lib.check(L.clapgpu_host_malloc(C.byref(h), n), "host_malloc")
C.memset(h, 3, n)
dst = np.empty(n, np.uint8)
src2 = np.full(n, 3, np.uint8)
for name, src in (("pinned", h.value), ("malloc", src2.ctypes.data)):
    for _ in range(2):
        t0 = time.perf_counter(); C.memmove(dst.ctypes.data, src, n); dt = time.perf_counter() - t0
    print(f"CPU memcpy 172 MB from {name}: {dt*1e3:.2f} ms {n/dt/1e9:.1f} GB/s")
L.clapgpu_host_free(h)
