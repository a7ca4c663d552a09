O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hostio_gpu.py tests/test_lod.py tests/test_physics_gpu.py -q -m gpu -x > $O/t.log 2>&1; echo "rc $?"; tail -25 $O/t.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
