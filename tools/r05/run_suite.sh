O=gpurun_out/r05
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/gpu_suite.log
tail -5 $O/gpu_suite.log
