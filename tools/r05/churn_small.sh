D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_hostio_gpu.py tests/test_dropin.py tests/test_scene_c.py tests/test_cabi.py -m gpu -x -q > $O/churn_small_tests.log 2>&1 || { tail -30 $O/churn_small_tests.log; exit 1; }
tail -2 $O/churn_small_tests.log
for rep in 1 2; do for args in "bench 10000 100 100 notify drawn churn 5" "bench 10000 100 100 notify drawn" "bench 10000 100 100 notify churn 5" "bench 100000 20 100 notify drawn churn 10" "bench 100000 20 100 notify drawn"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done > $O/churn_small.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/churn_small.log'):
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print(d['binding_mq_update_ms'], d['binding_ms'], d['reference_mq_update_ms'], d['mismatches'])
PY
