#!/bin/bash
# Same-session A/B of k_entities_tiles: the shipped library against another build of it (CLAPGPU_LIB), alternating,
# timed under rocprofv3 --kernel-trace at 1 M and 4 M entities.   tools/entities_ab.sh <outdir> <other libclapgpu.so>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/${1:-ent_ab}
other=${2:-$R/clap_amd/lib_ab/libclapgpu_old.so}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for chains in 125000 500000; do
for v in other shipped other shipped; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$other; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_$v" -- python3 "$R/tools/run_kernel.py" entities $chains 30 > "$out/$v.log" 2>&1
  f=$(find "$out/t_$v" -name '*kernel_stats.csv' | head -1)
  echo "$v $chains chains: $(grep k_entities_tiles "$f" | awk -F, '{printf "%.1f us (min %.1f)", $(NF-4)/1000, $(NF-2)/1000}')" | tee -a "$out/summary.txt"
  rm -rf "$out/t_$v"
done
done
