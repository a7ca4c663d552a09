#!/usr/bin/env python3
"""Fold tools/profile_entities_scale.sh's per-size files into <dir>/summary.json + a table on stdout."""
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
rows = {}
for path in sorted(glob.glob(os.path.join(d, "kernel_stats_*.csv"))):
    variant, chains = re.match(r"kernel_stats_(\w+?)_(\d+)\.csv", os.path.basename(path)).groups()
    for r in csv.DictReader(open(path)):
        if "k_entities_tiles" in r["Name"]:
            rows.setdefault((variant, int(chains)), {})["avg_us"] = float(r["AverageNs"]) / 1e3
            rows[(variant, int(chains))]["calls"] = int(r["Calls"])
for path in sorted(glob.glob(os.path.join(d, "entities_pmc_*.json"))):
    variant, chains = re.match(r"entities_pmc_(\w+?)_(\d+)\.json", os.path.basename(path)).groups()
    j = json.load(open(path))
    rows.setdefault((variant, int(chains)), {}).update(fetch_bytes=j["fetch_bytes_corrected_x2"], write_bytes=j["write_bytes"],
                                                       hbm_bytes=j["hbm_bytes_per_launch"])
out = []
for (variant, chains), r in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    n = chains * 8
    alg = 276 * (n - chains) + 212 * chains                       # 276 B per child, 212 B per root
    e = dict(variant=variant, entities=n, algorithmic_bytes=alg, **r)
    if "avg_us" in r:
        e["alg_frac_of_8TBs"] = alg / (r["avg_us"] * 1e-6) / 8e12
        if "hbm_bytes" in r:
            e["moved_frac_of_8TBs"] = r["hbm_bytes"] / (r["avg_us"] * 1e-6) / 8e12
    out.append(e)
    print(f"{variant:5s} {n:8d} entities  {r.get('avg_us', float('nan')):8.2f} us  alg {alg / 1e6:7.1f} MB "
          f"frac {e.get('alg_frac_of_8TBs', float('nan')):.3f}  moved {r.get('hbm_bytes', float('nan')) / 1e6:7.1f} MB "
          f"(r {r.get('fetch_bytes', float('nan')) / 1e6:.1f} + w {r.get('write_bytes', float('nan')) / 1e6:.1f}) "
          f"moved_frac {e.get('moved_frac_of_8TBs', float('nan')):.3f}")
json.dump(dict(note="k_entities_tiles<true>, chains x depth 8, all dirty; nt = shipped library (non-temporal stores), plain = "
                    "-DCLAPGPU_PLAIN_STORES A/B build; PMC: FETCH_SIZE x2 + WRITE_SIZE, separate passes (MI355X_MICROARCH.md)",
               rows=out), open(os.path.join(d, "summary.json"), "w"), indent=1)
