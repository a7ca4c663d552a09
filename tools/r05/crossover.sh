O=gpurun_out/r05
mkdir -p $O
timeout -k 10 1000 python3 tools/crossover.py > $O/crossover.json 2> $O/crossover.err; echo rc $?
tail -3 $O/crossover.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05/crossover.json'))
for k,v in d.items():
    if isinstance(v,dict):
        print(k, {x:v[x] for x in v if x!='rows'})
        for r in v['rows']: print('   ', r)
PY
