#!/bin/bash
# Entity kernel beyond the 256 MiB Infinity Cache, and the NT-store A/B the roofline `traffic` rests on.
#   tools/profile_entities_scale.sh r02_entities_scale
# For 1 M / 2 M / 4 M entities (125k / 250k / 500k chains x depth 8): rocprofv3 --kernel-trace --stats, then FETCH_SIZE and
# WRITE_SIZE in separate counter-only passes, for the shipped library (non-temporal stores) and for an A/B build with
# default-policy stores (clap_amd/lib_plain, make EXTRA="-DCLAPGPU_EXPERIMENT -DCLAPGPU_PLAIN_STORES" OUT=../lib_plain).  Output: gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r02_entities_scale}
out=$R/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for chains in 125000 250000 500000; do
  for variant in nt plain; do
    if [ $variant = plain ]; then
      [ -f "$R/clap_amd/lib_plain/libclapgpu.so" ] || continue
      export CLAPGPU_LIB=$R/clap_amd/lib_plain/libclapgpu.so
    else
      unset CLAPGPU_LIB
    fi
    t=$out/${variant}_$chains
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$t/trace" -- python3 "$R/tools/run_kernel.py" entities $chains 30 > "$t.log" 2>&1 || exit 1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$t/fetch" -- python3 "$R/tools/run_kernel.py" entities $chains 5 >> "$t.log" 2>&1 || exit 1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$t/write" -- python3 "$R/tools/run_kernel.py" entities $chains 5 >> "$t.log" 2>&1 || exit 1
    f=$(find "$t/trace" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "Name|clapgpu" "$f" > "$out/kernel_stats_${variant}_$chains.csv"
    fc=$(find "$t/fetch" -name '*counter_collection.csv' | head -1); wc=$(find "$t/write" -name '*counter_collection.csv' | head -1)
    [ -n "$fc" ] && [ -n "$wc" ] && python3 "$R/tools/pmc_summary.py" "$fc" "$wc" k_entities_tiles "$out/entities_pmc_${variant}_$chains.json" > /dev/null
    rm -rf "$t"
    echo "done $variant $chains"
  done
done
unset CLAPGPU_LIB
python3 "$R/tools/entities_scale_summary.py" "$out"
