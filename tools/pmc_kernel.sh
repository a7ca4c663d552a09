#!/bin/bash
# PMC pass over one secondary kernel on the GPU box: tools/pmc_kernel.sh <which> "<COUNTER ...>" [iters]
# (counters only, with --kernel-trace; never combined with the API/runtime trace domains)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
which=$1; counters=$2; iters=${3:-3}
out=$R/gpurun_out/pmc_${which}_$(echo $counters | tr ' ' '_' | cut -c1-40)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$out" -- python3 "$R/tools/run_kernel.py" "$which" "$iters" > "$out.log" 2>&1
f=$(find "$out" -name '*counter_collection.csv' | head -1)
if [ -z "$f" ]; then tail -5 "$out.log"; exit 1; fi
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, c) in sorted(acc.items()):
    if "at::" in kn or "rocclr" in kn: continue
    print(f"{kn:40s} {cn:24s} {s / c:16.1f}  (x{c})")
PY
