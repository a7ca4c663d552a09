#!/usr/bin/env python3
"""Run ONE of the secondary kernels a few times (for rocprofv3 --pmc / --kernel-trace).

    python tools/run_kernel.py pose|skin|particles|bodies|broadphase|contacts|lights [iters]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    import torch
    from clap_amd import _lib, animation, particles, physics, synth
    _lib.check(_lib.lib().clapgpu_init(0), "init")
    dev = "cuda:0"
    if which == "entities":                                     # entities <chains> <iters>: chains x depth 8, all dirty
        from clap_amd import entities, tiler
        chains, iters = iters, int(sys.argv[3]) if len(sys.argv) > 3 else 20
        scene = tiler.tiled_scene(synth.entities_chains(chains, 8, seed=2))[0]
        fr, _v, _p = entities.view_calc_frustum(synth.camera())
        batch = entities.EntityBatch(scene, dev)
        for _ in range(iters):
            batch.mq_update(fr, all_dirty=True)
            batch.compact_visible()
        torch.cuda.synchronize()
        print(f"{batch.n_real} entities, {batch.algorithmic_bytes()} algorithmic bytes per launch")
        return
    if which == "frame":                                        # one clap_frame() of everything, 23 times
        import bench_extras as bench
        print(bench.full_frame(dev)["ms_per_frame"], "ms per frame")
        return
    if which in ("pose", "skin"):
        J, n_chars, vpc = 64, 50_000, 200
        if len(sys.argv) > 4:                                   # pose|skin <iters> <characters> <vertices per character>
            n_chars, vpc = int(sys.argv[3]), int(sys.argv[4])
        sk = synth.skeleton(J, 8, seed=3)
        an = synth.animation(J, 30, 2.0, seed=3, missing_frac=float(os.environ.get("CLAP_POSE_MISSING", "0")))   # share of channels absent
        ch = synth.characters(n_chars, J, seed=3)
        mesh = synth.skinned_mesh(vpc, J, seed=3, copies=n_chars) if which == "skin" else None
        vf = (np.arange(n_chars, dtype=np.int64) * vpc).astype(np.uint32) if mesh else None
        vc = np.full(n_chars, vpc, np.uint32) if mesh else None
        model = animation.SkinnedModel(sk, [an], mesh=mesh, device=dev)
        cb = animation.CharacterBatch(model, n_chars, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
        cb.set_frame_times(ch["phase"])
        sk_ = int(os.environ.get("CLAP_POSE_SKIP", "0"))       # 1: no T/R/S write-back, 2: no joint positions
        cb.set_outputs(trs=not (sk_ & 1), joint_pos=not (sk_ & 2))
        cb.pose_update()
        fn = cb.pose_update if which == "pose" else cb.skin
        if which == "pose" and os.environ.get("CLAP_POSE_PLAYBACK"):      # animated_update at 60 Hz: the clock kernel moves every
            cb.start_clock(ani_time=-ch["phase"].astype(np.float64))     # character's frame time on by 1 / 60 s per launch
            clock = [0.0]
            def fn():
                clock[0] += 1.0 / 60.0
                cb.animated_update(clock[0])
    elif which == "particles":
        ps = synth.particle_systems(n_sys=4096, count=1024, radius=10.0, velocity=0.005, dist=synth.PART_DIST_SQRT)
        pos, vel, st = synth.particles_spawn(ps, synth.DRAND48_DEFAULT_STATE)
        pb = particles.ParticleBatch(ps, pos, vel, st, dev)
        view = np.eye(4, dtype=np.float32).ravel()
        fn = lambda: pb.particles_update(view)
    elif which == "lights":
        from clap_amd import lights as gl
        from clap_amd import entities as _ent
        ls = gl.LightSet(dev, 3840, 2160, 16)                    # 4K at 16-px tiles: 32 400 tiles
        ls.load(synth.lights(128, seed=7))
        _fr, vm, pm = _ent.view_calc_frustum(synth.camera(pos=(1.0, 2.0, 3.0)))
        fn = lambda: ls.grid_compute(vm, pm)
    else:
        if which.endswith("_spheres"):
            which = which[:-8]
            b = synth.sphere_bodies(262_144, box=64.0, seed=4)
        else:
            b = synth.capsule_bodies(262_144, box=60.0, seed=4)
        pw = physics.PhysWorld(b, synth.static_boxes(64, 60.0), pair_capacity=2_000_000, device=dev)
        pw.broadphase()
        fn = (lambda: pw.world_step(1 / 120)) if which == "bodies" else pw.contacts_geoms if which == "contacts" else pw.contacts_geoms_both if which == "contacts_both" else pw.broadphase
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
