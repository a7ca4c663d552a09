# a host-side race of the binding's worker passes shows in some runs only: keep the stderr of the runs that mismatch
D=oracle/_ref/clap_dropin
O=gpurun_out/r05/race; mkdir -p $O
bad=0
for i in $(seq 1 14); do
  timeout -k 10 120 $D test 200000 24 11 notify drawn comeandgo plain > $O/out_$i.txt 2> $O/err_$i.txt
  if grep -q '"mismatches": 0}' $O/out_$i.txt; then rm -f $O/out_$i.txt $O/err_$i.txt; else bad=$((bad+1)); echo "run $i: mismatch"; head -8 $O/err_$i.txt | cut -c1-1400; fi
done
echo "race hunt: $bad of 14 runs mismatched"
