D=oracle/_ref/clap_dropin
O=gpurun_out/r05
nproc > $O/nproc.txt; lscpu | head -20 >> $O/nproc.txt
for args in "test 300 12 1 notify drawn" "test 5000 24 2 notify drawn steady" "test 40000 12 3 notify drawn steady" "lod 3000 16 1 notify drawn steady" "lod 20000 10 2 notify" "lod 3000 12 1" "test 300000 6 5 notify drawn steady" "edge"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -4 | cut -c1-700
done > $O/dropin_drawn4.log 2>&1
for th in 8 16; do for args in "bench 1000000 6 1000 notify" "bench 1000000 6 1000 notify drawn" "bench 1000000 6 100 notify drawn" "bench 10000 30 100 notify drawn"; do
  echo "== threads $th $args"; GPU_SCENE_THREADS=$th GPU_SCENE_TIMING=1 timeout -k 10 300 $D $args 2>&1 | tail -3 | cut -c1-1700
done; done > $O/bench_drawn4.log 2>&1
