#!/bin/bash
# Same-session A/B of one kernel over several builds of the library, alternating, timed under rocprofv3 --kernel-trace.
#   tools/kernel_ab.sh <outdir> <lib1,lib2,...> <kernel substring> <run_kernel.py args...>
# A library is a path relative to the repository root, or "shipped" for clap_amd/lib/libclapgpu.so.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$1; libs=${2//,/ }; kern=$3; shift 3
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
for v in $libs; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$R/$v; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t" -- python3 "$R/tools/run_kernel.py" "$@" > "$out/last.log" 2>&1
  f=$(find "$out/t" -name '*kernel_trace.csv' | head -1)
  python3 - "$f" "$kern" "$v" <<'PY' | tee -a "$out/summary.txt"
import csv, sys, statistics
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = d[1:] if len(d) > 3 else d                      # the first launch pays the code upload
print(f'{sys.argv[3][-28:]:28s} {sys.argv[2]:16s} n {len(d):3d}  median {statistics.median(d):7.1f} us  mean {statistics.mean(d):7.1f}  min {min(d):7.1f}  max {max(d):7.1f}')
PY
  rm -rf "$out/t"
done
done
