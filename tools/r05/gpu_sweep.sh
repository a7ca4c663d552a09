# seeded sweep of the scripted game on the real device: entities coming and going in frames that are not walked
D=oracle/_ref/clap_dropin
O=gpurun_out/r05; mkdir -p $O
bad=0; n=0
: > $O/gpu_sweep.log
for seed in $(seq 300 339); do
  for cfg in "200 50" "1500 20" "12000 10"; do
    for pol in "notify drawn comeandgo plain" "notify comeandgo" "notify drawn comeandgo" "notify drawn steady" "" "steady" "comeandgo plain"; do
      set -- $cfg
      out=$(timeout -k 10 120 $D test $1 $2 $seed $pol 2>$O/gpu_sweep_err.txt | tail -1)
      n=$((n+1))
      if ! echo "$out" | grep -q '"mismatches": 0}'; then bad=$((bad+1)); echo "BAD seed=$seed cfg=$cfg pol=$pol: $out" >> $O/gpu_sweep.log; head -6 $O/gpu_sweep_err.txt | cut -c1-600 >> $O/gpu_sweep.log; fi
    done
  done
  echo "seed $seed done ($n runs, $bad bad)"
done
for seed in 1 2 3 4 5 6; do
  out=$(timeout -k 10 200 $D bench 50000 12 200 notify drawn churn $((seed * 17)) 2>/dev/null | tail -1)
  echo "$out" | grep -q '"mismatches": 0,' || { bad=$((bad+1)); echo "BAD bench churn $seed: $out" | cut -c1-600 >> $O/gpu_sweep.log; }
  n=$((n+1))
done
echo "gpu sweep done: $n runs, $bad bad" | tee -a $O/gpu_sweep.log
