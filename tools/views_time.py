#!/usr/bin/env python3
"""What a further view costs inside the entity update's launch: BASELINE configs[1] (1 M entities, all dirty) with 0, 1, 2 and 4
extra frusta culled by the same launch (k_entities_tiles<true> / k_entities_tiles_xv), against a separate k_entities_cull pass
per view.      python tools/views_time.py [launches]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    import torch
    from clap_amd import _lib, entities, synth, tiler
    _lib.check(_lib.lib().clapgpu_init(0), "init")
    scene = tiler.tiled_scene(synth.entities_chains(125_000, 8, seed=2))[0]
    fr, _v, _p = entities.view_calc_frustum(synth.camera())
    import math
    frusta = []
    for k in range(4):                                          # four "light" views: other places, turned about Y, narrower and shorter
        a = 0.35 * (k + 1)
        cam = synth.camera(pos=(40.0 * k - 60.0, 20.0, 60.0), quat=(0.0, math.sin(a / 2), 0.0, math.cos(a / 2)), fov_deg=50.0, aspect=1.0,
                           near=1.0, far=300.0)
        frusta.append(entities.view_calc_frustum(cam)[0])
    batch = entities.EntityBatch(scene, "cuda:0")

    def timed(fn):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters

    out = {}
    for n in (0, 1, 2, 4):
        batch.set_views(frusta[:n])
        out[f"update_with_{n}_extra_views_us"] = round(timed(lambda: batch.mq_update(fr, all_dirty=True)), 2)
    batch.set_views([])
    out["separate_cull_pass_us"] = round(timed(lambda: batch.cull(frusta[0])), 2)
    print(out)


if __name__ == "__main__":
    main()
