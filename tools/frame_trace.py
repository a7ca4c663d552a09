#!/usr/bin/env python3
"""A few BASELINE-size frames for `rocprofv3 --kernel-trace`: which kernels of the frame's three chains run at the same time?
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/frame_trace -- python3 tools/frame_trace.py [frames] [serial]
tools/frame_trace_summary.py turns the trace into per-frame timelines."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench_extras as bench
    from clap_amd import _lib
    _lib.check(_lib.lib().clapgpu_init(0), "init")
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
    orig = bench.time_launches
    calls = [0]

    def few(fn, iters, warmup=3):
        calls[0] += 1
        if calls[0] == 1:
            return orig(fn, frames, warmup=40)
        return 1.0                                              # the serial / graph legs of bench.full_frame: skipped
    bench.time_launches = few
    if serial:
        from clap_amd import frame
        frame.FrameLoop.overlap = False
    else:
        from clap_amd import frame
        frame.FrameLoop.overlap = True
    r = bench.full_frame("cuda:0")
    print(f"{r['ms_per_frame']:.3f} ms per frame over {frames} frames ({'one stream' if serial else 'three chains'})")


if __name__ == "__main__":
    main()
