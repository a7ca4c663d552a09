#!/bin/bash
# Time one secondary kernel under rocprofv3 on the GPU box:  tools/prof_kernel.sh pose|skin|particles|bodies|broadphase [iters]
# Prints the kernel_stats rows; the CSVs land in gpurun_out/prof_<name>/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
which=$1; iters=${2:-20}
out=$R/gpurun_out/prof_$which
cd /tmp && export TMPDIR=/tmp
rm -rf "$out"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$R/tools/run_kernel.py" "$which" "$iters" > "$out.log" 2>&1
rm -f /dev/null; f=$(find "$out" -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cut -d, -f1-8 "$f" | grep -v -E "at::|elementwise" | head -12; else tail -5 "$out.log"; fi
