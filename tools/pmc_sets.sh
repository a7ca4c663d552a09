#!/bin/bash
# Instruction / wait / cache counters of the kernels matching <regex>, from six counters-only rocprofv3 passes over
# `python3 <script> <args...>`:   tools/pmc_sets.sh <tag> <regex> <script> [args...]   ->  gpurun_out/<tag>/pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; re=$2; shift 2
out=$R/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -- python3 "$@" > "$out/p$i.log" 2>&1
done
python3 - "$out" "$re" > "$out/pmc.txt" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
rx = re.compile(sys.argv[2])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            acc[r["Kernel_Name"].split("(")[0].replace("clapgpu::", "").replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:36s} mean {sum(v)/len(v):14.1f}   (n={len(v)})")
PY
rm -rf "$out"/p[0-9]
cat "$out/pmc.txt"
