O=gpurun_out/r05
mkdir -p $O
for v in shipped b1024_w4_l2 b768_w3_l2 b768_w4 b1024_w4_t3200 shipped b1024_w4_l2 b768_w3_l2 b768_w4 b1024_w4_t3200; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$PWD/clap_amd/lib_ab/$v/libclapgpu.so; fi
  echo "== $v"
  CLAPGPU_POSE_DEBUG=1 timeout -k 10 200 python3 tools/pose_time.py 150 2>&1 | grep -v "^$" | sort | uniq -c | sort -rn | head -6 | cut -c1-300
done > $O/pose_ab.log 2>&1
cat $O/pose_ab.log
