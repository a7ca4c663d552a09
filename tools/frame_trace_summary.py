#!/usr/bin/env python3
"""Timeline of the LAST frames of a rocprofv3 --kernel-trace CSV of tools/frame_trace.py: per kernel its start / end relative
to the frame's first kernel and the queue it ran on, and how much of the frame had two or more kernels in flight.
    python tools/frame_trace_summary.py <kernel_trace.csv> [frames_to_show]"""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].replace("clapgpu::", "").replace("void ", "")
        if "at::" in name or "elementwise" in name or "rocclr" in name.lower():
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:44], r.get("Queue_Id", "?")))
    rows.sort()
    # frames start with the broadphase's first kernel (chain A) or the animation clock (chain B), whichever comes first
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_animation_time")]
    show = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for fi in starts[-show - 1:-1]:
        nxt = starts[starts.index(fi) + 1]
        fr = rows[fi:nxt]
        t0 = min(r[0] for r in fr)
        t1 = max(r[1] for r in fr)
        ev = sorted([(r[0], 1) for r in fr] + [(r[1], -1) for r in fr])
        live, last, busy1, busy2 = 0, t0, 0, 0
        for t, d in ev:
            if live >= 1: busy1 += t - last
            if live >= 2: busy2 += t - last
            live += d
            last = t
        print(f"frame of {(t1 - t0) / 1e3:.1f} us: {busy1 / 1e3:.1f} us with a kernel in flight, {busy2 / 1e3:.1f} us with two or more; "
              f"sum of kernel durations {sum(r[1] - r[0] for r in fr) / 1e3:.1f} us")
        for r in fr:
            print(f"  {(r[0] - t0) / 1e3:8.1f} .. {(r[1] - t0) / 1e3:8.1f} us  q{r[3]:>3s}  {r[2]}")


if __name__ == "__main__":
    main()
