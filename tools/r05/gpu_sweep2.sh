# seeded sweep on the device with the worker passes forced from a few hundred entities (and at their defaults for big queues)
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
bad=0; n=0
: > $O/gpu_sweep2.log
for seed in $(seq 800 829); do
  for cfg in "2000 20" "17000 10" "70000 6"; do
    for pol in "notify drawn comeandgo plain" "notify comeandgo" "notify drawn steady" "steady"; do
      set -- $cfg
      out=$(GPU_SCENE_MIRROR_PAR_MIN=300 GPU_SCENE_SCATTER_PAR_MIN=200 timeout -k 10 120 $D test $1 $2 $seed $pol 2>$O/gpu_sweep2_err.txt | tail -1)
      n=$((n+1))
      if ! echo "$out" | grep -q '"mismatches": 0}'; then bad=$((bad+1)); echo "BAD seed=$seed cfg=$cfg pol=$pol: $out" | cut -c1-500 >> $O/gpu_sweep2.log; head -4 $O/gpu_sweep2_err.txt | cut -c1-600 >> $O/gpu_sweep2.log; fi
    done
  done
  echo "seed $seed done ($n runs, $bad bad)"
done
for seed in 1 2 3 4; do
  out=$(timeout -k 10 200 $D test 250000 16 $((900 + seed)) notify drawn comeandgo plain 2>/dev/null | tail -1); n=$((n+1))
  echo "$out" | grep -q '"mismatches": 0}' || { bad=$((bad+1)); echo "BAD big $seed: $out" | cut -c1-500 >> $O/gpu_sweep2.log; }
done
echo "gpu sweep 2 done: $n runs, $bad bad" | tee -a $O/gpu_sweep2.log
