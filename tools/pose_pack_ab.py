#!/usr/bin/env python3
"""k_pose at BASELINE configs[2] (50 000 characters x 64 joints), key-major pools (clapgpu_animations_pack) against the
channel-major pools, alternating in one process: medians of HIP-event-timed launches.
python tools/pose_pack_ab.py [rounds] [joints] [characters]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import torch
    from clap_amd import _lib, animation, synth
    _lib.check(_lib.lib().clapgpu_init(0), "clapgpu_init")
    J = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    sk = synth.skeleton(J, 8 if J <= 64 else 12, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    batches = {}
    with_mesh = os.environ.get("POSE_AB_MESH") == "1"            # as bench.py's extras: a distinct 200-vertex mesh per character
    mesh = synth.skinned_mesh(200, J, seed=3, copies=n) if with_mesh else None
    vf = (np.arange(n, dtype=np.int64) * 200).astype(np.uint32) if with_mesh else None
    vc = np.full(n, 200, np.uint32) if with_mesh else None
    for name, pack in (("key_major", True), ("channel_major", False)):
        model = animation.SkinnedModel(sk, [an], mesh=mesh, device="cuda:0", pack=pack)
        cb = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
        cb.set_frame_times(ch["phase"])
        batches[name] = cb

    def timed(cb, iters=300, warm=100):
        for _ in range(warm):
            cb.pose_update()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); cb.pose_update(); b.record()
        torch.cuda.synchronize()
        t = np.asarray([a.elapsed_time(b) for a, b in ev]) * 1e3
        return float(np.median(t)), float(t.min()), float(t.mean())

    out = {k: [] for k in batches}
    for mode in ("all outputs", "palette only"):
        for cb in batches.values():
            cb.set_outputs(trs=mode == "all outputs", joint_pos=mode == "all outputs")
        for _ in range(rounds):
            for name, cb in batches.items():
                out[name].append((mode, *timed(cb)))
    res = {k: [dict(mode=m, median_us=a, best_us=b, mean_us=c) for m, a, b, c in v] for k, v in out.items()}
    res["joints"], res["characters"] = J, n
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
