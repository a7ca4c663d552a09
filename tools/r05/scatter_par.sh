D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for thr in 12288 32768 131072; do for args in "bench 70000 30 100" "bench 100000 20 100 notify" "bench 100000 20 300 notify" "bench 300000 10 100 notify drawn" "bench 1000000 5 100 notify drawn"; do
  echo "== thr=$thr $args"; GPU_SCENE_SCATTER_PAR_MIN=$thr timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done; done > $O/scatter_par.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/scatter_par.log'):
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print(d['binding_mq_update_ms'], d['binding_ms']['scatter'], d['mismatches'])
PY
