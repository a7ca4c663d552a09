#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/ent_valu; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in shipped x1 x2; do
  if [ $v = shipped ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$R/clap_amd/lib_$v/libclapgpu.so; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$out/p_$v" -- python3 "$R/tools/run_kernel.py" entities 125000 5 > "$out/$v.log" 2>&1
  f=$(find "$out/p_$v" -name '*counter_collection.csv' | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_entities_tiles" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
  rm -rf "$out/p_$v"
done
