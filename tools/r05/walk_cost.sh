# what a WALKED frame costs now (the fallback): every frame forced through the walk + re-tile by GPU_SCENE_INCREMENTAL=0
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for args in "bench 1000000 5 100 notify drawn churn 10" "bench 1000000 5 100 notify churn 10" "bench 100000 10 100 notify drawn churn 10" "bench 1000000 4 100"; do
  echo "== incremental=0 $args"; GPU_SCENE_INCREMENTAL=0 timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/walk_cost.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/walk_cost.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','fast_frames','retiles','mismatches')})
PY
