#!/usr/bin/env python3
"""Write a scene snapshot (include/clapgpu_snapshot.h) of one of the BASELINE workloads.

    python tools/make_snapshot.py c2 out.clps        # 1M entities, 125k chains x depth 8 + camera
    python tools/make_snapshot.py c1 out.clps        # 10k flat entities (the CPU-reference case)
    python bench.py --snapshot out.clps
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clap_amd import snapshot, synth  # noqa: E402


def main():
    which, path = sys.argv[1], sys.argv[2]
    if which == "c1":
        ents = synth.entities_flat(10_000, seed=1234)
    elif which == "c2":
        ents = synth.entities_chains(125_000, 8, seed=2)
    else:
        raise SystemExit("c1 | c2")
    snapshot.save_scene(path, entities=ents, camera=synth.camera())
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
