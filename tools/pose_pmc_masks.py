#!/usr/bin/env python3
"""k_pose's HBM bytes PER OUTPUT MASK into a round's pmc_hbm_bytes.json.

bench.py launches k_pose with all outputs AND palette-only (clapgpu_pose_batch.skip) under one kernel name and grid, so
the all-kernels summary (tools/pmc_summary.py --all) can only give their mixed mean.  This tool takes four counters-only
passes of `tools/run_kernel.py pose` -- FETCH_SIZE and WRITE_SIZE, each with CLAP_POSE_SKIP=0 (all outputs) and =3
(palette only) -- and writes one entry per mask, replacing the mixed one:
    "<kernel> [all outputs]", "<kernel> [palette only]"
    python tools/pose_pmc_masks.py <fetch_all.csv> <write_all.csv> <fetch_pal.csv> <write_pal.csv> <pmc_hbm_bytes.json>
Units and the gfx950 FETCH_SIZE correction as in tools/pmc_summary.py (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys


def mean(path, counter):
    vals, name = [], None
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_pose<" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not vals:
        raise SystemExit(f"no {counter} rows of k_pose in {path}")
    return sum(vals) / len(vals), len(vals), name


def main():
    fa, wa, fp, wp, out = sys.argv[1:6]
    try:
        res = json.load(open(out))
    except (OSError, ValueError):
        res = dict(note="", kernels={})
    for label, f, w in (("all outputs", fa, wa), ("palette only", fp, wp)):
        f_kib, nf, name = mean(f, "FETCH_SIZE")
        w_kib, nw, _ = mean(w, "WRITE_SIZE")
        res["kernels"][f"{name} [{label}]"] = dict(dispatches=nf, FETCH_SIZE_mean_kib_raw=f_kib, WRITE_SIZE_mean_kib=w_kib,
                                                   fetch_bytes_corrected_x2=f_kib * 2048, write_bytes=w_kib * 1024,
                                                   hbm_bytes_per_launch=f_kib * 2048 + w_kib * 1024)
    for k in [k for k in res["kernels"] if "k_pose<" in k and "[" not in k]:
        del res["kernels"][k]                                # the mixed mean of both masks: not a number of any launch
    res["note"] = (res.get("note", "") + "  k_pose per output mask: four passes of tools/run_kernel.py pose (CLAP_POSE_SKIP=0 / 3), "
                   "tools/pose_pmc_masks.py").strip()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in res["kernels"].items() if "k_pose<" in k}, indent=1))


if __name__ == "__main__":
    main()
