#!/usr/bin/env python3
"""How much of k_pose's output is BIT-identical to the oracle's (the reference's arithmetic on the host)?
    python tools/pose_exact_check.py [n_chars] [frames]
Prints, per output array, the number of objects (joints) with any differing bit and the worst difference in ulps."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clap_amd import synth, animation                     # noqa: E402
from oracle import binding as ob                          # noqa: E402


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    sk = synth.skeleton(64, 8, seed=3)
    an = synth.animation(64, 30, 2.0, seed=3)
    ch = synth.characters(n, 64, seed=3)
    sk["bind"] = ob.skeleton_bind(sk)
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device="cuda:0")
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    trs = np.tile(ch["trs0"], (n, 1, 1))
    for f in range(frames):
        t = ((ch["phase"] + 0.37 * f) % 2.1 - 0.02).astype(np.float32)
        jt, gl, jp = ob.pose(sk, an, t, ch["char_mx"], trs)
        batch.set_frame_times(t)
        batch.pose_update()
        out = batch.download()
        for name, got, exp in (("T", out["trs"][..., 0:3], trs[..., 0:3]), ("R", out["trs"][..., 3:7], trs[..., 3:7]), ("S", out["trs"][..., 7:10], trs[..., 7:10]),
                               ("joint_transforms", out["joint_transforms"], jt), ("joint_pos", out["joint_pos"], jp)):
            got = np.ascontiguousarray(got, np.float32).reshape(n * 64, -1); exp = np.ascontiguousarray(exp, np.float32).reshape(n * 64, -1)
            neq = got != exp                              # value comparison: -0 == +0
            u = ulps(got, exp) * neq
            bad = neq.any(axis=1)
            rel = np.abs(got.astype(np.float64) - exp).max(axis=1) / np.maximum(np.abs(exp).max(axis=1), 1e-30)
            if name == "R" and bad.any():                # which of the interpolation's branches: |a.b| of the differing joints' result vs unit length
                k = np.flatnonzero(bad)[:6]
                print("   R samples (got | exp):", [(got[i].tolist(), exp[i].tolist()) for i in k[:3]])
            print(f"frame {f} {name:17s}: {int(bad.sum()):8d} of {bad.size} joints differ, worst {int(u.max())} ulp, "
                  f"worst relative to the object {rel.max():.3e}, objects over 1e-5: {int((rel > 1e-5).sum())}")


if __name__ == "__main__":
    main()
