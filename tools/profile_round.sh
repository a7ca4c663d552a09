#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun): kernel-trace stats of bench.py, then the two
# PMC passes (FETCH_SIZE, WRITE_SIZE: separately, as MI355X_MICROARCH.md prescribes; counters only,
# with --kernel-trace) and an unprofiled bench line.  Output: gpurun_out/<tag>/ ; copy what should be
# judged into profiles/<tag>/.
#     tools/profile_round.sh r01_final
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r01_final}
out=$R/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$R/bench.py" --steps 100 --warmup 10 --cpu-frames 0 --no-testbed > "$out/bench_under_rocprof.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-frames 0 --no-testbed > "$out/bench_fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-frames 0 --no-testbed > "$out/bench_write.log" 2>&1
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
# the headline kernel alone (no extras, no testbed frame, no drop-in scenes): every k_entities_tiles launch is BASELINE size
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_h" -- python3 "$R/bench.py" --steps 200 --warmup 20 --cpu-frames 0 --no-testbed --no-extras > "$out/bench_headline_under_rocprof.log" 2>&1
f=$(find "$out/trace_h" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats_headline.csv"
fc=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
wc=$(find "$out/write" -name '*counter_collection.csv' | head -1)
if [ -n "$fc" ] && [ -n "$wc" ]; then
    python3 "$R/tools/pmc_summary.py" --all "$fc" "$wc" "$out/pmc_hbm_bytes.json" > /dev/null
fi
# k_pose per OUTPUT MASK (bench.py launches both masks under one kernel name and grid: the summary above can only mix them)
for m in 0 3; do for c in FETCH_SIZE WRITE_SIZE; do
    CLAP_POSE_SKIP=$m timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pose_${c}_$m" -- python3 "$R/tools/run_kernel.py" pose 8 > "$out/pose_${c}_$m.log" 2>&1
done; done
pf() { find "$out/pose_$1_$2" -name '*counter_collection.csv' | head -1; }
if [ -n "$(pf FETCH_SIZE 0)" ] && [ -n "$(pf WRITE_SIZE 0)" ] && [ -n "$(pf FETCH_SIZE 3)" ] && [ -n "$(pf WRITE_SIZE 3)" ]; then
    python3 "$R/tools/pose_pmc_masks.py" "$(pf FETCH_SIZE 0)" "$(pf WRITE_SIZE 0)" "$(pf FETCH_SIZE 3)" "$(pf WRITE_SIZE 3)" "$out/pmc_hbm_bytes.json" > "$out/pose_pmc_masks.txt"
fi
rm -rf "$out"/pose_FETCH_SIZE_* "$out"/pose_WRITE_SIZE_*
# the headline kernel's traffic from headline-only passes: in the full bench the whole-frame extra launches the same grid
# with only part of the scene dirty, which would pull the per-launch mean down
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch_h" -- python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-frames 0 --no-testbed --no-extras > "$out/bench_fetch_headline.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write_h" -- python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-frames 0 --no-testbed --no-extras > "$out/bench_write_headline.log" 2>&1
fch=$(find "$out/fetch_h" -name '*counter_collection.csv' | head -1)
wch=$(find "$out/write_h" -name '*counter_collection.csv' | head -1)
[ -n "$fch" ] && [ -n "$wch" ] && python3 "$R/tools/pmc_summary.py" "$fch" "$wch" k_entities_tiles "$out/entities_pmc.json" > /dev/null
rm -rf "$out/trace" "$out/trace_h" "$out/fetch" "$out/write" "$out/fetch_h" "$out/write_h"           # keep the summaries, not the per-dispatch dumps
ls -la "$out"; tail -c 600 "$out/bench_unprofiled.json"; echo; [ -f "$out/kernel_stats.csv" ] && cut -d, -f1-4 "$out/kernel_stats.csv" | cut -c1-110 | head -40
