#!/usr/bin/env python3
"""From which scene size on is the binding faster than the reference's own host loop?  Sweeps oracle/_ref/clap_dropin
(the reference's objects on both sides, same box, reference first) over entities, animated characters and particle
systems and prints one JSON document: per size both times, and per family the smallest size from which the binding wins
and keeps winning.  Needs a GPU.      python tools/crossover.py > profiles/r05_experiments/crossover.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "clap_dropin")


def run(*args):
    p = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        raise RuntimeError(f"clap_dropin {args}: rc {p.returncode}: {p.stderr[-400:]}")
    return json.loads(p.stdout.strip().splitlines()[-1])


def crossover(rows, key_ref, key_bind):
    """smallest n such that the binding is at least as fast there and at every larger n measured"""
    win = None
    for r in sorted(rows, key=lambda r: r["n"], reverse=True):
        if r[key_bind] <= r[key_ref]:
            win = r["n"]
        else:
            break
    return win


def main():
    out = {}
    # ---- entities: mq_update alone, and the frame a maintainer gets (mutators + mq_update + draw list)
    for permille in (100, 1000):
        rows = []
        for n in (100, 300, 1000, 3000, 10_000, 30_000, 100_000):
            frames = 200 if n <= 3000 else (60 if n <= 30_000 else 12)
            r = run("bench", n, frames, permille, "notify")
            rows.append(dict(n=n, reference_mq_update_ms=r["reference_mq_update_ms"], binding_mq_update_ms=r["binding_mq_update_ms"],
                             reference_frame_ms=r["reference_frame_ms"], binding_frame_draw_list_ms=r["binding_frame_draw_list_ms"],
                             identical=r["mismatches"] == 0 and r["draw_reads_equal"]))
        out[f"entities_{permille // 10}pct_moving"] = dict(
            rows=rows, mq_update_wins_from=crossover(rows, "reference_mq_update_ms", "binding_mq_update_ms"),
            frame_wins_from=crossover(rows, "reference_frame_ms", "binding_frame_draw_list_ms"))
    # ---- animated characters x 64 joints (mq_update + animated_update against gpu_mq_update + gpu_anim_update)
    rows = []
    for n in (1, 2, 5, 10, 15, 20, 30, 50, 100, 200, 500):
        frames = 300 if n <= 50 else 60
        r = run("anim", n, 64, frames, 5, "notify")
        rows.append(dict(n=n, reference_ms=r["reference_ms_per_frame"], binding_ms=r["binding_ms_per_frame"],
                         identical=r["mismatches"] == 0 and r.get("differing_objects") == 0))
    out["characters_64_joints"] = dict(rows=rows, wins_from=crossover(rows, "reference_ms", "binding_ms"))
    # ---- particle systems x 512 particles
    rows = []
    for n in (1, 2, 4, 8, 12, 16, 20, 32, 64, 128):
        r = run("particles", n, 512, 200 if n <= 32 else 60, 4)
        rows.append(dict(n=n, particles=n * 512, reference_ms=r["reference_ms_per_frame"], binding_ms=r["binding_ms_per_frame_positions_only"],
                         identical=r["mismatches"] == 0 and r["stream_draws_agree"]))
    out["particle_systems_x_512"] = dict(rows=rows, wins_from=crossover(rows, "reference_ms", "binding_ms"))
    out["note"] = ("wins_from: the smallest measured size from which the binding is at least as fast as the reference's host loop AND "
                   "stays so at every larger size measured (null: never, in the range).  Below it the fixed cost of a device round trip "
                   "(launch, start, completion word over PCIe: ~20 us around a few-us kernel) is more than the whole host loop.")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
