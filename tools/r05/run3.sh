D=oracle/_ref/clap_dropin
O=gpurun_out/r05
( DROPIN_TRACE=285 timeout -k 10 300 $D test 300 12 1 notify drawn 2>&1 | cut -c1-900
  DROPIN_TRACE=4237 timeout -k 10 300 $D test 40000 12 3 notify drawn steady 2>&1 | grep "trace\|entity 4237" | cut -c1-900 ) > $O/trace.log 2>&1
