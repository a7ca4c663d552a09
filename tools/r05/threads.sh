D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for th in 16 24 0 16 24 0; do
  echo "== threads $th"; if [ $th = 0 ]; then unset GPU_SCENE_THREADS; else export GPU_SCENE_THREADS=$th; fi
  GPU_SCENE_TIMING=1 timeout -k 10 300 $D bench 1000000 8 1000 notify drawn 2>&1 | tail -3 | cut -c1-900
done > $O/threads.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/threads.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d[k] for k in ('binding_mq_update_ms','binding_ms','binding_mutate_ms','binding_draw_list_ms','binding_frame_draw_list_ms','mismatches')})
PY
