# kernel-level view of frames with in-place edits: the one-launch host frame + k_entities_place
O=gpurun_out/r05/prof_churn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o churn -- $GRAFT_REPO_ROOT/oracle/_ref/clap_dropin bench 100000 40 100 notify drawn churn 10 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_churn.csv
head -12 $O/kernel_stats_churn.csv | cut -c1-200
tail -1 $O/run.log | cut -c1-400
