# the other modes of the checker, longer and bigger than the test suite runs them
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for args in "anim 2000 64 60 5 notify" "anim 300 200 40 6" "anim 20000 40 10 7 notify" "characters 30000 30 2 notify" "characters 2000 120 3" "particles 256 1024 40 3" "particles 2000 512 10 4" "lights 300 3" "recreate 300000" "edge" "snapshot 50000 gpurun_out/r05/soak_snapshot.bin" "lod 300000 10 16 notify drawn steady" "lod 50000 30 17 comeandgo plain" "fail 50 20000 12 3 notify drawn comeandgo plain"; do
  echo "== $args"; timeout -k 10 500 $D $args > $O/soak2_out.txt 2> $O/soak2_err.txt; echo "rc $?"; tail -1 $O/soak2_out.txt | cut -c1-600; grep -i "mismatch\|differ" $O/soak2_err.txt | head -3 | cut -c1-300
done > $O/soak2.log 2>&1
rm -f gpurun_out/r05/soak_snapshot.bin
grep -c "^== " $O/soak2.log; grep "^rc" $O/soak2.log | sort | uniq -c; python3 - <<'PY'
import json
for l in open('gpurun_out/r05/soak2.log'):
    if l.startswith('=='): print(l.strip(), end='  ')
    elif l.startswith('{'):
        try:
            d=json.loads(l); print({k:d.get(k) for k in ('mismatches','differing_objects','joint_attached_differing','cases')})
        except Exception as e: print('unparsed', l[:120])
PY
