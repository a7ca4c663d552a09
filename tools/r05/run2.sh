D=oracle/_ref/clap_dropin
O=gpurun_out/r05
for args in "test 300 12 1 notify drawn" "test 300 24 1 notify drawn steady" "test 5000 24 2 notify drawn steady" "test 40000 12 3 notify drawn steady" "lod 3000 16 1 notify drawn steady" "test 5000 24 2 notify steady"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -12 | cut -c1-1200
done > $O/dropin_drawn2.log 2>&1
for args in "test 300000 6 5 notify drawn steady" "test 300000 6 5 notify steady" "bench 10000 30 100 notify" "bench 10000 30 100 notify drawn" "bench 1000000 6 1000 notify" "bench 1000000 6 1000 notify drawn" "bench 1000000 6 100 notify drawn"; do
  echo "== $args"; GPU_SCENE_TIMING=1 timeout -k 10 300 $D $args 2>&1 | tail -8 | cut -c1-1700
done > $O/bench_drawn2.log 2>&1
