#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<round>_entities_pmc.json.

Usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <out.json>

HBM bytes per launch are derived as MI355X_MICROARCH.md (section HBM) prescribes:
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one), both
are in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced
stream at 64 B, i.e. reports exactly half the bytes -> doubled here.  WRITE_SIZE is exact
for 16-B-per-lane streaming stores.
"""
import collections
import csv
import json
import sys


def mean_kib(path, needle, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if needle in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if not vals:
        raise SystemExit(f"no {counter} rows for '{needle}' in {path}")
    return sum(vals) / len(vals), len(vals)


def all_kernels(fetch_csv, write_csv, out):
    """Every clapgpu kernel of the two passes -> {kernel: corrected bytes per launch}."""
    # bench.py launches most kernels at several sizes (BASELINE size, the testbed-sized frame, the drop-in scenes): only
    # the dispatches with the kernel's LARGEST grid -- the BASELINE-size launches -- are averaged.
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path, counter in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "clapgpu" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][counter].append(
                    (int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    kernels = {}
    for name, c in sorted(acc.items()):
        if not c["FETCH_SIZE"] or not c["WRITE_SIZE"]:
            continue
        grid = max(g for g, _ in c["FETCH_SIZE"])
        fv = [v for g, v in c["FETCH_SIZE"] if g == grid]
        wv = [v for g, v in c["WRITE_SIZE"] if g == grid]
        if not wv:
            continue
        f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
        kernels[name] = dict(dispatches=len(fv), dispatches_all_sizes=len(c["FETCH_SIZE"]), grid_size=grid,
                             FETCH_SIZE_mean_kib_raw=f_kib, WRITE_SIZE_mean_kib=w_kib,
                             fetch_bytes_corrected_x2=f_kib * 2048, write_bytes=w_kib * 1024,
                             hbm_bytes_per_launch=f_kib * 2048 + w_kib * 1024)
    res = dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 10 "
                    "--warmup 2`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); per kernel, the dispatches with its largest grid (the BASELINE-size launches) only",
               kernels=kernels)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def main():
    if sys.argv[1] == "--all":
        return all_kernels(*sys.argv[2:5])
    fetch_csv, write_csv, needle, out = sys.argv[1:5]
    f_kib, nf = mean_kib(fetch_csv, needle, "FETCH_SIZE")
    w_kib, nw = mean_kib(write_csv, needle, "WRITE_SIZE")
    fetch_bytes = f_kib * 1024 * 2          # gfx950 correction (see docstring)
    write_bytes = w_kib * 1024
    res = dict(kernel=needle, dispatches_fetch_pass=nf, dispatches_write_pass=nw,
               FETCH_SIZE_mean_kib_raw=f_kib, WRITE_SIZE_mean_kib=w_kib,
               fetch_bytes_corrected_x2=fetch_bytes, write_bytes=write_bytes,
               hbm_bytes_per_launch=fetch_bytes + write_bytes,
               note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
