#!/usr/bin/env python3
"""Time the full BASELINE-size frame (bench.full_frame's scene) in consecutive batches, to see warm-up effects."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench_extras as bench
    from clap_amd import _lib
    _lib.check(_lib.lib().clapgpu_init(0), "init")
    orig = bench.time_launches
    res = []

    def probe(fn, iters, warmup=3):
        for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
            res.append(orig(fn, 20, warmup=0) * 1e3)
        return res[-1] * 1e-3
    bench.time_launches = probe
    bench.full_frame("cuda:0")
    print(" ".join(f"{r:.3f}" for r in res), "ms per frame, consecutive batches of 20")


if __name__ == "__main__":
    main()
