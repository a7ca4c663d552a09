O=gpurun_out/r05
mkdir -p $O
timeout -k 10 1100 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"
