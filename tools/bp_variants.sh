#!/bin/bash
# tools/bp_variants.sh <variant...>: tools/bp_time.py under each clap_amd/lib_ab/<variant>/libclapgpu.so (experiment builds) and the shipped library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in base "$@"; do
  if [ "$v" = base ]; then unset CLAPGPU_LIB; else export CLAPGPU_LIB=$R/clap_amd/lib_ab/$v/libclapgpu.so; fi
  echo "== $v"
  CLAPGPU_ALLOW_EXPERIMENT=1 timeout -k 10 120 python3 "$R/tools/bp_time.py" 2>&1 | sed 's/contacts us.*//'
done
