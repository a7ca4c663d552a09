D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for args in "bench 1000000 5 100 notify drawn churn 10" "bench 1000000 5 100 notify drawn" "bench 100000 10 100 notify drawn churn 10" "bench 100000 10 100 notify drawn" "bench 10000 30 100 notify drawn churn 5" "bench 10000 30 100 notify drawn"; do
  echo "== $args"; GPU_SCENE_TIMING=1 CLAPGPU_SCENE_TIMING=1 timeout -k 10 300 $D $args 2>&1 | tail -4 | cut -c1-1000
done > $O/churn.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/churn.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d[k] for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','binding_frame_draw_list_ms','reference_frame_ms','mismatches')})
PY
