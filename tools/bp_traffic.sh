#!/bin/bash
# HBM-side bytes of the broadphase kernels (tools/bp_time.py <kind>): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes,
# corrected as MI355X_MICROARCH.md prescribes (KiB; FETCH_SIZE x2 on gfx950).  tools/bp_traffic.sh <tag> [kind] -> gpurun_out/<tag>/traffic.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-bp_traffic}; kind=${2:-capsules}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -- python3 "$R/tools/bp_time.py" $kind > "$out/$c.log" 2>&1
done
python3 - "$out" $kind > "$out/traffic.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{sys.argv[1]}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("k_bp_" in r["Kernel_Name"] or "k_bodies_step" in r["Kernel_Name"] or "k_contacts" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("clapgpu::", "")][c].append(float(r["Counter_Value"]))
print("==", sys.argv[2], "(MB per launch: FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB)")
tot = 0.0
for k in sorted(acc):
    f = sum(acc[k]["FETCH_SIZE"]) / max(len(acc[k]["FETCH_SIZE"]), 1) * 2048 / 1e6
    w = sum(acc[k]["WRITE_SIZE"]) / max(len(acc[k]["WRITE_SIZE"]), 1) * 1024 / 1e6
    print(f"{k:44s} read {f:8.1f}  write {w:8.1f}  total {f + w:8.1f}   (n={len(acc[k]['FETCH_SIZE'])})")
    if "k_bp_" in k: tot += f + w
print(f"{'sum of the k_bp_* kernels':44s} {tot:8.1f}")
PY
rm -rf "$out/FETCH_SIZE" "$out/WRITE_SIZE"
cat "$out/traffic.txt"
