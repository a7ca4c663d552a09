# what the room a re-tile leaves for in-place edits (an eighth of every row, a spare row per tile) costs frames WITHOUT edits
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
for inc in 1 0; do
for args in "bench 1000000 5 1000 notify drawn" "bench 1000000 5 100 notify drawn" "bench 10000 40 1000 notify" "bench 10000 40 100 notify drawn"; do
  echo "== incremental=$inc $args"; GPU_SCENE_INCREMENTAL=$inc timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done; done > $O/slack_ab.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/slack_ab.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','binding_mutate_ms','binding_draw_list_ms','binding_frame_draw_list_ms','reference_frame_ms','mismatches')})
PY
