# in-place creation / deletion: the mirror against the oracle, the drop-in checker's games, then the churn bench with the
# edits taken in place and (GPU_SCENE_INCREMENTAL=0) through a walk + re-tile as before
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_scene_c.py tests/test_dropin.py -m gpu -x -q > $O/churn2_tests.log 2>&1 || { tail -30 $O/churn2_tests.log; exit 1; }
tail -3 $O/churn2_tests.log
for t in "test 2500 16 1 notify drawn comeandgo" "test 300 60 5 notify drawn comeandgo" "test 2000 40 7 notify comeandgo"; do
  echo "== $t"; timeout -k 10 300 $D $t 2>&1 | tail -2 | cut -c1-900
done > $O/churn2_games.log 2>&1
cat $O/churn2_games.log
for inc in 1 0; do
for args in "bench 1000000 5 100 notify drawn churn 10" "bench 1000000 5 100 notify drawn churn 100" "bench 100000 10 100 notify drawn churn 10" "bench 10000 30 100 notify drawn churn 5" "bench 10000 30 100 notify drawn"; do
  echo "== incremental=$inc $args"; GPU_SCENE_INCREMENTAL=$inc timeout -k 10 300 $D $args 2>&1 | tail -3 | cut -c1-1600
done; done > $O/churn2.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/churn2.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','binding_frame_draw_list_ms','reference_frame_ms','fast_frames','retiles','placed_in_layout','removed_in_place','mismatches')})
    else: print(l.strip()[:300])
PY
