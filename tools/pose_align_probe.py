#!/usr/bin/env python3
"""Does k_pose's launch time depend on WHERE its three output arrays lie relative to each other?  (Its all-outputs time
is 88-110 us from one process to the next while palette-only is stable at 78 us.)  One big allocation; joint_transforms
at its start, T/R/S and joint positions behind it with swept paddings; median of 200 launches each.
python tools/pose_align_probe.py > gpurun_out/pose_align.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clap_amd import _lib, animation, synth
    _lib.check(_lib.lib().clapgpu_init(0), "clapgpu_init")
    J, n = 64, 50_000
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    model = animation.SkinnedModel(sk, [an], device="cuda:0")
    cb = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    cb.set_frame_times(ch["phase"])
    jt_b, trs_b, jp_b = n * J * 64, n * J * 40, n * J * 16
    big = torch.zeros((jt_b + trs_b + jp_b + (64 << 20)) // 4, dtype=torch.float32, device="cuda:0")
    base = big.data_ptr()
    base_al = (base + (1 << 21) - 1) & ~((1 << 21) - 1)               # 2 MiB aligned start
    off0 = (base_al - base) // 4
    trs0 = cb.trs.clone()

    def view(byte_off, nbytes, shape):
        a = off0 + byte_off // 4
        return big[a:a + nbytes // 4].view(*shape)

    def timed(iters=200, warm=60):
        for _ in range(warm):
            cb.pose_update()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); cb.pose_update(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]) * 1e3)

    out = []
    if len(sys.argv) > 1 and sys.argv[1] == "fine":                    # T/R/S placement in 64 KiB steps over 4 MiB, then the positions'
        sweep = [(p1, 0) for p1 in range(0, 4 << 20, 64 << 10)]
    elif len(sys.argv) > 2 and sys.argv[1] == "pos":
        sweep = [(int(sys.argv[2]), p2) for p2 in range(0, 4 << 20, 128 << 10)]
    else:
        sweep = [(p1, p2) for p1 in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, (1 << 21) + 65536 + 256)
                 for p2 in (0, 4096, 65536 + 1024)]
    for p1, p2 in sweep:
        if True:
            o_jt = 0
            o_trs = (jt_b + p1 + 255) & ~255
            o_jp = (o_trs + trs_b + p2 + 255) & ~255
            jt = view(o_jt, jt_b, (n, J, 16)); trs = view(o_trs, trs_b, (n, J, 10)); jp = view(o_jp, jp_b, (n, J, 4))
            trs.copy_(trs0)
            cb.joint_transforms, cb.trs, cb.joint_pos = jt, trs, jp
            cb._pose_desc.joint_transforms, cb._pose_desc.trs, cb._pose_desc.joint_pos = jt.data_ptr(), trs.data_ptr(), jp.data_ptr()
            out.append(dict(pad_trs=p1, pad_pos=p2, trs_minus_jt_mod_2M=(o_trs - o_jt) % (1 << 21), us=timed()))
            print(out[-1], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
