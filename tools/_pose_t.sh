cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pp -- python3 $GRAFT_REPO_ROOT/tools/run_kernel.py pose 30 > /dev/null 2>&1; f=$(find $GRAFT_REPO_ROOT/gpurun_out/pp -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_pose' in r['Name']: print('$1', r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, 'us')
"; rm -rf $GRAFT_REPO_ROOT/gpurun_out/pp; }
CLAPGPU_POSE_ONE_WAVE=1 run "one-wave skip=0"
CLAPGPU_POSE_ONE_WAVE=1 CLAP_POSE_SKIP=1 run "one-wave skip=1"
CLAPGPU_POSE_ONE_WAVE=1 CLAP_POSE_SKIP=3 run "one-wave skip=3"
run "pc skip=0"
