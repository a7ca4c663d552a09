R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05/frame_prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CLAP_FRAME_ONE_STREAM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/frame_probe.py 6 > $out/frame_probe.log 2>&1
f=$(find $out/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_frame.csv
rm -rf $out/trace
cat $out/frame_probe.log | tail -2
cut -d, -f1-6 $out/kernel_stats_frame.csv | cut -c1-140 | head -40
