# the worker pool's wake-up: futex on the generation word vs condition variable + mutex (two builds of the checker)
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do for v in futex cv; do for args in "bench 20000 80 1000 notify" "bench 30000 60 300 notify drawn" "bench 60000 40 100 notify" "bench 40000 40 100" "bench 100000 20 100 notify drawn" "bench 1000000 5 100 notify drawn"; do
  echo "== $v $args"; timeout -k 10 300 oracle/_ref/clap_dropin_$v $args 2>&1 | tail -1 | cut -c1-1700
done; done; done > $O/pool_ab.log 2>&1
python3 - <<'PY'
import json, collections
acc=collections.defaultdict(list)
key=None
for l in open('gpurun_out/r05/pool_ab.log'):
    if l.startswith('=='): key=l.strip()[3:]
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        acc[key].append((d['binding_mq_update_ms'], d['mismatches']))
for k,v in acc.items(): print(k, [x[0] for x in v], 'mismatches', sum(x[1] for x in v))
PY
