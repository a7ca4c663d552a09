O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests_b.log 2>&1; rc=$?; tail -4 $O/gpu_tests_b.log; [ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_b.json 2> $O/bench_b.err; rc=$?; tail -c 3000 $O/bench_b.json; exit $rc
