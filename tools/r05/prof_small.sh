# kernel-level view of a testbed-sized notified frame (10 k entities): how much of the device step is the kernel
O=gpurun_out/r05/prof_small; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for args in "bench 10000 200 100 notify drawn" "bench 10000 200 1000 notify"; do
  tag=$(echo $args | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag -o t -- $GRAFT_REPO_ROOT/oracle/_ref/clap_dropin $args > $GRAFT_REPO_ROOT/$O/$tag.log 2>&1
  echo "== $args"; head -5 $GRAFT_REPO_ROOT/$O/$tag/t_kernel_stats.csv | cut -c1-60,150-260
  tail -1 $GRAFT_REPO_ROOT/$O/$tag.log | cut -c1-330
done
