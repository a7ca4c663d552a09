# the one-launch frame also brings over what earlier frames left stale and this one reads: no second launch for "came into view"
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_hostio_gpu.py tests/test_dropin.py tests/test_scene_c.py -m gpu -x -q > $O/late_tests.log 2>&1 || { tail -30 $O/late_tests.log; exit 1; }
tail -2 $O/late_tests.log
for args in "bench 10000 200 100 notify drawn" "bench 10000 200 1000 notify drawn" "bench 100000 20 100 notify drawn" "bench 1000000 5 1000 notify drawn" "bench 1000000 5 100 notify drawn churn 10"; do
  echo "== $args"; timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700
done > $O/late.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/late.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l[:l.index(', "note"')]+'}') if ', "note"' in l else json.loads(l)
        print({k:d.get(k) for k in ('reference_mq_update_ms','binding_mq_update_ms','binding_ms','binding_frame_draw_list_ms','reference_frame_ms','fetched_on_view_per_frame','left_stale_per_frame','mismatches')})
PY
