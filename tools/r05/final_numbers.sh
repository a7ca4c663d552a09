# the boundary rows of README / DESIGN on one box, final tree
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
for args in "bench 1000000 5 1000 notify drawn" "bench 1000000 5 1000 notify" "bench 1000000 8 100 notify drawn" "bench 1000000 8 100 notify drawn churn 10" "bench 1000000 8 100 notify drawn churn 100" "bench 1000000 5 100" "bench 1000000 5 1000" \
            "bench 100000 20 100 notify drawn" "bench 100000 20 100 notify drawn churn 10" "bench 100000 20 100" \
            "bench 10000 100 100 notify drawn" "bench 10000 100 100 notify" "bench 10000 100 1000 notify" "bench 10000 100 100 notify drawn churn 5" "bench 10000 100 100"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/final_numbers.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/final_numbers.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print('   ', {k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','reference_frame_ms','binding_frame_draw_list_ms','binding_mutate_ms','binding_draw_list_ms','mismatches')})
PY
