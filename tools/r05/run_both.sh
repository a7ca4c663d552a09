O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?"
tail -3 $O/gpu_suite.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
