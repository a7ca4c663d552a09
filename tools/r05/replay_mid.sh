D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for rp in 1 0; do for args in "bench 10000 50 100" "bench 16500 50 100" "bench 20000 50 100" "bench 40000 40 100" "bench 64000 30 100" "bench 70000 30 100"; do
  echo "== replay=$rp $args"; GPU_SCENE_REPLAY=$rp timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done; done > $O/replay_mid.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/replay_mid.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','frames_by_the_records','mismatches')})
PY
