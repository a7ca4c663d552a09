#!/usr/bin/env python3
"""One clap_frame() of a testbed-sized scene (BASELINE configs[0] scale: 10k flat entities, a few characters,
particle systems and bodies): where the frame is launch latency, not bytes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench_extras as bench
    from clap_amd import _lib, animation, characters, entities, frame, lights, particles, physics, synth, tiler
    from oracle import binding as ob
    _lib.check(_lib.lib().clapgpu_init(0), "init")
    dev = "cuda:0"
    raw = synth.entities_flat(10_000, seed=1234)
    scene, tl = tiler.tiled_scene(raw)
    roots = tl["slot_of"][np.flatnonzero(raw["parent"] < 0)]
    batch = entities.EntityBatch(scene, dev)
    n_bodies, n_chars, J, vpc = 128, 10, 64, 2000
    b = synth.sphere_bodies(n_bodies, box=16.0, seed=4)
    b["body_entity"] = roots[:n_bodies].astype(np.int32)
    world = physics.PhysWorld(b, synth.static_boxes(16, 16.0), pair_capacity=4096, device=dev)
    feed = synth.character_feed(n_chars, seed=13, with_bodies=False)
    feed["entity"] = roots[n_bodies:n_bodies + n_chars].astype(np.uint32)
    cf = characters.CharacterFeed(feed, dev)
    ls = lights.LightSet(dev, 1920, 1080, lights.TILE_WIDTH)
    ls.load(synth.lights(16, seed=7))
    ls.set_carriers(roots[-8:].astype(np.uint32), np.arange(8, dtype=np.int32), np.zeros((8, 3), np.float32))
    sk, an = synth.skeleton(J, 8, seed=3), synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n_chars, J, seed=3)
    mesh = synth.skinned_mesh(vpc, J, seed=3)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, device=dev)
    cb = animation.CharacterBatch(model, n_chars, ch["trs0"], batch.mx, entity_index=feed["entity"],
                                  vert_first=np.zeros(n_chars, np.uint32), vert_count=np.full(n_chars, vpc, np.uint32))
    cb.start_clock(ani_time=-ch["phase"].astype(np.float64), speed=np.ones(n_chars, np.float32))
    ps = synth.particle_systems(n_sys=8, count=1024, radius=10.0, velocity=0.005)
    pos, vel, st = ob.particles_spawn(ps, synth.DRAND48_DEFAULT_STATE)
    pb = particles.ParticleBatch(ps, pos, vel, st, dev)
    loop = frame.FrameLoop(batch, synth.camera(), world=world, feed=cf, lights=ls, characters=cb, particles=pb, contacts=True)
    now = [0.0]

    def one():
        now[0] += 1.0 / 120.0
        loop.clap_frame(now[0], 1.0 / 120.0)
    t = bench.time_launches(one, 200, warmup=100)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        one()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    print(f"small frame: {t * 1e6:.1f} us on the device stream, {wall * 1e6:.1f} us wall per frame (host-side launches included)")
    # the same frame as one captured HIP graph
    loop.capture(1.0 / 120.0, warmup_now=now[0])

    def replay():
        now[0] += 1.0 / 120.0
        loop.clap_frame_replay(now[0])
    tg = bench.time_launches(replay, 200, warmup=100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        replay()
    torch.cuda.synchronize()
    wall_g = (time.perf_counter() - t0) / 200
    print(f"small frame, graph replay: {tg * 1e6:.1f} us on the device stream, {wall_g * 1e6:.1f} us wall per frame")


if __name__ == "__main__":
    main()
