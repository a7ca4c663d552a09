#!/bin/bash
# Per-kernel hashes of the gfx950 assembly of one .hip file (comments, directives and labels' metadata stripped): the
# check that a change elsewhere did not move a measured kernel.  Runs without a GPU.
#     tools/isa_hashes.sh clap_amd/csrc/entities.hip > profiles/r03_d/entities_isa_hashes.txt
f=${1:?usage: tools/isa_hashes.sh file.hip}
d=$(cd "$(dirname "$f")" && pwd); R=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp /tmp/isa_XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I "$R/include" -I "$d" -Wall -Wno-unused-function \
    --offload-device-only -S "$f" -o "$tmp" $EXTRA 2> >(grep -v "hip-link" >&2) || exit 1
python3 - "$tmp" <<'PY'
import sys, re, hashlib
cur, body = None, {}
for l in open(sys.argv[1]).read().splitlines():
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1); body[cur] = []; continue
    if cur is not None:
        if l.startswith('.Lfunc_end'):
            cur = None; continue
        s = l.split(';')[0].rstrip()
        if s.strip() and not s.strip().startswith('.'):
            # block labels carry the function's ordinal in the translation unit (.LBB7_3): adding a kernel elsewhere
            # in the file renumbers them without moving an instruction
            body[cur].append(re.sub(r'\.LBB\d+_', '.LBB_', s))
for k, v in sorted(body.items()):
    print(hashlib.md5('\n'.join(v).encode()).hexdigest()[:12], len(v), k)
PY
rm -f "$tmp"
