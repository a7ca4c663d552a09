# long runs of the in-place paths on the device: growth tiles, capacity exhaustion -> walk + re-tile -> in place again
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for args in "bench 1000000 30 100 notify drawn churn 300" "bench 200000 60 100 notify churn 500" "bench 20000 300 100 notify drawn churn 60" "test 200000 24 11 notify drawn comeandgo plain" "test 60000 60 12 notify comeandgo plain" "test 3000 400 13 notify drawn comeandgo plain" "test 100000 30 14 steady" "lod 100000 16 15 notify drawn comeandgo plain"; do
  echo "== $args"; timeout -k 10 500 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/soak.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/soak.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print('   ', {k:d.get(k) for k in ('mismatches','fast_frames','frames_by_the_records','retiles','placed_in_layout','removed_in_place','binding_mq_update_ms','reference_mq_update_ms','draw_sets_equal')})
    else: print(l.strip()[:300])
PY
