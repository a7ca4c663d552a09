# churn with the lazily-made room + the same queue without edits (packed tight)
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_dropin.py tests/test_scene_c.py -m gpu -x -q -k "coming_and_going or not_walked or bench_mode or scene_mirror or abi" > $O/churn3_tests.log 2>&1 || { tail -30 $O/churn3_tests.log; exit 1; }
tail -2 $O/churn3_tests.log
for args in "bench 1000000 8 100 notify drawn churn 10" "bench 1000000 8 100 notify drawn churn 100" "bench 1000000 8 1000 notify drawn churn 10" "bench 1000000 5 1000 notify drawn" "bench 100000 20 100 notify drawn churn 10" "bench 100000 20 100 notify drawn" "bench 10000 60 100 notify drawn churn 5" "bench 10000 60 100 notify drawn" "bench 10000 60 100 notify churn 5"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/churn3.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/churn3.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','binding_frame_draw_list_ms','reference_frame_ms','fast_frames','retiles','placed_in_layout','removed_in_place','mismatches')})
PY
