#!/bin/bash
# PMC passes over tools/bp_time.py, summarised per broadphase kernel: gpurun_out/<tag>/pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-bp_pmc}
KIND=${2:-}
out=$R/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -- python3 "$R/tools/bp_time.py" $KIND > "$out/p$i.log" 2>&1
done
python3 - "$out" > "$out/pmc.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bp_" in r["Kernel_Name"] or "k_contacts" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("clapgpu::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:36s} mean {sum(v)/len(v):14.1f}   (n={len(v)})")
PY
rm -rf "$out"/p[0-9]
cat "$out/pmc.txt"
