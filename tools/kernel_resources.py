#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel, from the compiler's own metadata (hipcc -S of each .hip with the
Makefile's flags; cross-compiles without a GPU).      python tools/kernel_resources.py > profiles/<round>/kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clap_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-function", "--cuda-device-only", "-S"]
PER_FILE = {"contacts.hip": ["-mllvm", "-simplifycfg-sink-common=false"]}     # as in the Makefile


def main():
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch B':>9s} {'LDS B':>7s}")
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip"):
                continue
            out = os.path.join(tmp, f + ".s")
            p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *PER_FILE.get(f, []), os.path.join(CSRC, f), "-o", out], capture_output=True, text=True)
            if p.returncode:
                print(f"{f}: {p.stderr[-300:]}", file=sys.stderr)
                continue
            t = open(out).read()
            rows = []
            for m in re.finditer(r"- \.agpr_count.*?\.wavefront_size", t, re.S):
                b = m.group(0)
                g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name).replace("void ", "").replace("clapgpu::", "")
                rows.append((name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                             g("private_segment_fixed_size"), g("group_segment_fixed_size")))
            if rows:
                print(f"# {f}")
            for r in sorted(rows):
                print(f"{r[0][:70]:70s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>6s} {r[5]:>6s} {r[6]:>9s} {r[7]:>7s}")


if __name__ == "__main__":
    main()
