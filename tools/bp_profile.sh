#!/bin/bash
# per-kernel times of the broadphase at C4 size (tools/bp_time.py under rocprofv3): gpurun_out/<tag>/bp_kernels.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-bp}
out=$R/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
: > "$out/bp_kernels.txt"; : > "$out/bp_time.log"
for kind in spheres capsules; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$kind" -- python3 "$R/tools/bp_time.py" $kind >> "$out/bp_time.log" 2>&1
f=$(find "$out/trace_$kind" -name '*kernel_stats.csv' | head -1)
python3 - "$f" $kind >> "$out/bp_kernels.txt" <<'PY'
import csv, sys
print("==", sys.argv[2])
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if "clapgpu" in r["Name"]:
        print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
        if "k_bp_" in r["Name"]: tot += float(r['AverageNs'])/1e3
print(f"{'sum of the k_bp_* kernels':60s}       {tot:9.1f} us")
PY
rm -rf "$out/trace_$kind"
done
grep "broadphase us" "$out/bp_time.log"; cat "$out/bp_kernels.txt"
