O=gpurun_out/r05
mkdir -p $O
( time timeout -k 10 1100 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc $?"
tail -3 $O/bench_time.txt; tail -5 $O/bench_default.err
