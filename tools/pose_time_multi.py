#!/usr/bin/env python3
"""k_pose on skeletons of more than 64 joints (2-4 wavefronts per character) and with missing channels, 3.2 M joints each:
    python tools/pose_time_multi.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                              # noqa: E402
from clap_amd import synth, animation                     # noqa: E402


def main():
    for J, n, depth, akw in ((64, 50_000, 8, {}), (64, 50_000, 8, dict(missing_frac=0.1)), (128, 25_000, 10, {}), (192, 16_667, 14, {}),
                             (200, 16_000, 16, {}), (64, 50_000, 8, dict(keyframes=60))):
        sk = synth.skeleton(J, depth, seed=3)
        an = synth.animation(J, akw.pop("keyframes", 30), 2.0, seed=3, **akw)
        ch = synth.characters(n, J, seed=3)
        model = animation.SkinnedModel(sk, [an], device="cuda:0")
        cb = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
        cb.set_frame_times(ch["phase"])
        for _ in range(30):
            cb.pose_update()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            cb.pose_update()
        e1.record()
        torch.cuda.synchronize()
        print(f"{J:4d} joints x {n:6d} characters {akw or ''}: {e0.elapsed_time(e1) * 1000 / 50:.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
