#!/bin/bash
# k_pose's kernel time for small batches (the binding's testbed-sized frames): rocprofv3 --kernel-trace per batch size.
#   tools/pose_small.sh <tag> [sizes...]      -> gpurun_out/<tag>/pose_small.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
sizes=${@:-10 200 2000}
out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for n in $sizes; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_$n" -- python3 "$R/tools/run_kernel.py" pose 20 $n 1 > "$out/run_$n.log" 2>&1 || { echo "n=$n failed" >> "$out/pose_small.txt"; exit 1; }
  f=$(find "$out/t_$n" -name '*kernel_stats.csv' | head -1)
  echo "n=$n $( [ -n "$f" ] && grep k_pose "$f" < /dev/null | cut -d, -f1-8 | cut -c1-40,60-)" >> "$out/pose_small.txt"
  [ -n "$f" ] && grep k_pose "$f" >> "$out/pose_small_raw.txt"
done
cat "$out/pose_small.txt"
