# frames WITHOUT notifications (every frame walks the queue) and walked frames with a re-tile
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_dropin.py -m gpu -x -q > $O/walk2_tests.log 2>&1 || { tail -30 $O/walk2_tests.log; exit 1; }
tail -2 $O/walk2_tests.log
for args in "bench 1000000 4 100" "bench 1000000 4 1000" "bench 100000 10 100" "bench 10000 50 100" "bench 10000 50 1000"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/walk_cost2.log 2>&1
for args in "bench 1000000 5 100 notify drawn churn 10" "bench 100000 10 100 notify drawn churn 10"; do
  echo "== incremental=0 $args"; GPU_SCENE_INCREMENTAL=0 timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done >> $O/walk_cost2.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/walk_cost2.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','reference_frame_ms','binding_frame_draw_list_ms','fast_frames','retiles','mismatches')})
PY
