set -x
D=oracle/_ref/clap_dropin
O=gpurun_out/r05
timeout -k 10 300 python -m pytest tests/test_hostio_gpu.py tests/test_scene_c.py -x -q -m gpu > $O/t_hostio.log 2>&1; echo "hostio rc $?" 
for args in "test 300 12 1 notify" "test 300 12 1 notify drawn" "test 5000 16 2 notify drawn" "test 40000 8 3 notify drawn" "test 2000 40 7 notify drawn" "test 300000 6 5 notify drawn" "test 5000 16 2 drawn" "lod 3000 12 1 notify" "lod 3000 12 1 notify drawn" "lod 20000 10 2 notify drawn" "edge"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -12
done > $O/dropin_drawn.log 2>&1
for args in "bench 10000 30 100 notify" "bench 10000 30 100 notify drawn" "bench 10000 30 1000 notify" "bench 10000 30 1000 notify drawn" "bench 1000000 6 1000 notify" "bench 1000000 6 1000 notify drawn" "bench 1000000 6 100 notify" "bench 1000000 6 100 notify drawn"; do
  echo "== $args"; GPU_SCENE_TIMING=1 timeout -k 10 300 $D $args 2>&1 | tail -12
done > $O/bench_drawn.log 2>&1
tail -3 $O/t_hostio.log
