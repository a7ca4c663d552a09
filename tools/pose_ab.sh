#!/bin/bash
# Same-session A/B of k_pose over several builds of the library by HIP events (tools/pose_time.py, 3 x <iters> launches per
# output mask; the later repetitions are the settled ones), alternating, twice.
#   tools/pose_ab.sh <lib1,lib2,...> [iters]      a library: path relative to the repository root, or "shipped"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
libs=${1//,/ }; iters=${2:-150}
for round in 1 2; do
for v in $libs; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$R/$v; fi
  echo "== $v"
  timeout -k 10 200 python3 "$R/tools/pose_time.py" $iters </dev/null 2>&1 | grep k_pose || exit 1
done
done
