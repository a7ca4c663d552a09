#!/bin/bash
# A/B of the drop-in checker's bench / test modes on the GPU box, every variant x every scene, alternating, repeated:
#     tools/ab.sh <log name> [-r reps] [-e "VAR=a" -e "VAR=b" ...] [-t] -- "<clap_dropin args>" ["<args>" ...]
#   -e  one environment setting per variant (several -e = the variants that alternate; none = one plain variant);
#       "VAR=a OTHER=b" sets two variables for that variant
#   -t  also print the per-pass timing lines (GPU_SCENE_TIMING=1 CLAPGPU_SCENE_TIMING=1) into the log
# Log: gpurun_out/<log name>.log (copy what should be judged into profiles/); stdout: one summary line per run.
# What round 5's one-off scripts did, e.g.:
#     tools/ab.sh r06/churn -e GPU_SCENE_INCREMENTAL=1 -e GPU_SCENE_INCREMENTAL=0 -- "bench 1000000 5 100 notify drawn churn 10" "bench 100000 10 100 notify drawn churn 10"
#     tools/ab.sh r06/replay -r 3 -e GPU_SCENE_REPLAY_MIN=0 -e GPU_SCENE_REPLAY_MIN=16384 -- "bench 10000 400 100"
#     tools/ab.sh r06/retile -t -e GPU_SCENE_RETILE_BY_MASK=1 -e GPU_SCENE_RETILE_BY_MASK=0 -- "bench 1000000 5 100 notify churn 10"   (with GPU_SCENE_INCREMENTAL=0 exported)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/oracle/_ref/clap_dropin
name=$1; shift
reps=1; timing=0; variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case $1 in
    -r) reps=$2; shift 2 ;;
    -e) variants+=("$2"); shift 2 ;;
    -t) timing=1; shift ;;
    *) echo "tools/ab.sh: unknown option $1" >&2; exit 2 ;;
  esac
done
shift
[ ${#variants[@]} -eq 0 ] && variants=("")
log=$R/gpurun_out/$name.log; mkdir -p "$(dirname "$log")"; : > "$log"
for rep in $(seq 1 $reps); do for v in "${variants[@]}"; do for args in "$@"; do
  echo "== [$v] $args" >> "$log"
  if [ $timing = 1 ]; then
    env $v GPU_SCENE_TIMING=1 CLAPGPU_SCENE_TIMING=1 timeout -k 10 300 $D $args 2>&1 | grep -v "^scene small" | tail -8 | cut -c1-1700 >> "$log"
  else
    env $v timeout -k 10 300 $D $args 2>&1 | tail -1 | cut -c1-1700 >> "$log"
  fi
done; done; done
python3 - "$log" <<'PY'
import json, sys
key = None
for l in open(sys.argv[1]):
    if l.startswith('=='): key = l.strip()[3:]
    elif l.startswith('{'):
        try: d = json.loads(l[:l.index(', "note"')] + '}') if ', "note"' in l else json.loads(l)
        except ValueError:
            i = l.find(', "reference_mutate'); d = json.loads(l[:i] + '}') if i > 0 else {}
        keep = ('reference_mq_update_ms', 'binding_mq_update_ms', 'binding_ms', 'reference_frame_ms', 'binding_frame_draw_list_ms',
                'frames_by_the_records', 'fast_frames', 'retiles', 'mismatches')
        print(key, {k: d[k] for k in keep if k in d})
PY
