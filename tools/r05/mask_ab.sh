D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for bm in 8 4 2 1; do for args in "bench 1000000 5 1000 notify drawn" "bench 1000000 5 500 notify drawn" "bench 1000000 5 200 notify" "bench 1000000 5 1000 notify"; do
  echo "== sparse_factor=$bm $args"; GPU_SCENE_SCATTER_BY_MASK=$bm timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done; done > $O/mask_ab2.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/mask_ab2.log'):
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print(d['binding_mq_update_ms'], d['binding_ms']['scatter'], d['mismatches'])
PY
