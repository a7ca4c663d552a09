#!/bin/bash
# Is k_entities_tiles bound by instruction issue?  The same launch (same bytes in and out) with parts of the arithmetic
# compiled out (wrong results; A/B builds made by `make -C clap_amd/csrc OUT=../lib_x1 EXTRA="-DCLAPGPU_EXPERIMENT -DCLAPGPU_EXP_NO_AABB"` and
# `OUT=../lib_x2 EXTRA="-DCLAPGPU_EXPERIMENT -DCLAPGPU_EXP_NO_AABB -DCLAPGPU_EXP_NO_INVERT"`), timed under rocprofv3 at 1 M and 4 M entities.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/${1:-ent_sens}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for chains in 125000 500000; do
for v in shipped x1 x2; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$R/clap_amd/lib_$v/libclapgpu.so; [ -f "$CLAPGPU_LIB" ] || continue; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t_$v" -- python3 "$R/tools/run_kernel.py" entities $chains 30 > "$out/$v.log" 2>&1
  f=$(find "$out/t_$v" -name '*kernel_stats.csv' | head -1)
  echo "$v $chains chains: $(grep k_entities_tiles "$f" | awk -F, '{printf "%.1f us (min %.1f)", $(NF-4)/1000, $(NF-2)/1000}')" | tee -a "$out/summary.txt"
  rm -rf "$out/t_$v"
done
done
