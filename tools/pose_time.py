#!/usr/bin/env python3
"""k_pose at BASELINE configs[2] (50 000 characters x 64 joints): the mean launch time over K launches between one event
pair, for every output mask.    python tools/pose_time.py [iters]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                              # noqa: E402
from clap_amd import synth, animation                     # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    sk = synth.skeleton(64, 8, seed=3)
    an = synth.animation(64, 30, 2.0, seed=3)
    ch = synth.characters(50_000, 64, seed=3)
    model = animation.SkinnedModel(sk, [an], device="cuda:0")
    cb = animation.CharacterBatch(model, 50_000, ch["trs0"], ch["char_mx"])
    cb.set_frame_times(ch["phase"])
    for label, kw in (("all outputs", dict(trs=True, joint_pos=True)), ("palette only", dict(trs=False, joint_pos=False))):
        cb.set_outputs(**kw)
        for _ in range(5):
            cb.pose_update()
        torch.cuda.synchronize()
        best = []
        for _rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                cb.pose_update()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1000.0 / iters)
        print(f"k_pose {label}: " + " / ".join(f"{b:.1f}" for b in best) + " us per launch")


if __name__ == "__main__":
    main()
