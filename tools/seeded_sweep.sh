#!/bin/bash
# Seeded sweep of the drop-in checker's scripted game on the GPU box (round 5's gpu_sweep*.sh): every seed x scene size x
# policy set must end with 0 mismatches.   tools/seeded_sweep.sh <log name> <first seed> <last seed> [forced]
#   forced: the worker passes from a few hundred entities on (GPU_SCENE_MIRROR_PAR_MIN=300 GPU_SCENE_SCATTER_PAR_MIN=200)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/oracle/_ref/clap_dropin
log=$R/gpurun_out/$1.log; mkdir -p "$(dirname "$log")"; : > "$log"
forced=""; [ "$4" = forced ] && forced="GPU_SCENE_MIRROR_PAR_MIN=300 GPU_SCENE_SCATTER_PAR_MIN=200"
bad=0; n=0
for seed in $(seq $2 $3); do for cfg in "200 50" "1500 20" "12000 10"; do
  for pol in "notify drawn comeandgo plain" "notify comeandgo" "notify drawn comeandgo" "notify drawn steady" "" "steady" "comeandgo plain"; do
    set -- $cfg
    out=$(env $forced timeout -k 10 120 $D test $1 $2 $seed $pol 2>/tmp/sweep_err.txt | tail -1); n=$((n+1))
    if ! echo "$out" | grep -q '"mismatches": 0}'; then
      bad=$((bad+1)); echo "BAD seed=$seed cfg=$cfg pol=$pol: $out" | cut -c1-500 >> "$log"; head -6 /tmp/sweep_err.txt | cut -c1-600 >> "$log"
    fi
  done
done; done
echo "$n runs, $bad bad" | tee -a "$log"
[ $bad -eq 0 ]
