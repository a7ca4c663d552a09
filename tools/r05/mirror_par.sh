D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for thr in 16384 65536; do for args in "bench 20000 50 1000 notify" "bench 30000 40 1000 notify" "bench 50000 30 1000 notify" "bench 100000 20 200 notify" "bench 100000 20 300 notify drawn"; do
  echo "== thr=$thr $args"; GPU_SCENE_MIRROR_PAR_MIN=$thr timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done; done > $O/mirror_par.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/mirror_par.log'):
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print(d['binding_mq_update_ms'], d['binding_ms']['walk'], d['binding_ms']['scatter'], d['reference_mq_update_ms'], d['mismatches'])
PY
