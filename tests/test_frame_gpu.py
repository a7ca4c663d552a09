"""End to end: several display frames of the whole batched path in the reference's frame order
(clap_amd.frame.FrameLoop) against the same sequence run on the oracle -- rigid bodies, characters
with bodies and position history, light carriers, rotation pushes, the entity hierarchy, animation
clock + pose + skinning, particles, cull + LOD, light grid."""
import numpy as np
import pytest

from clap_amd import _lib, synth
from oracle import binding as ob
from helpers import bits_equal

pytestmark = pytest.mark.gpu
E_DIRTY = 1 << 16


@pytest.mark.parametrize("prebin", [False, True], ids=["plain", "prebin"])
def test_frames_match_oracle_sequence(prebin, cuda_device):
    """prebin: CLAPGPU_FRAME_PREBIN -- every substep's body step also bins its boxes for the next broadphase pass."""
    import torch
    from clap_amd import animation, characters, entities, frame, lights, particles, physics

    n_dyn, n_char, J, vpc = 200, 100, 24, 60
    # ---- entities: [dynamic-body roots | character roots | a forest of props], level-major
    forest = synth.entities_forest(1500, seed=12, n_models=3, max_depth=4)
    roots = np.flatnonzero(forest["parent"] < 0)
    assert len(roots) >= n_dyn + n_char + 20
    scene = synth.pad_levels(forest)
    slot = scene["slot_of"]
    dyn_e, char_e, prop_e = slot[roots[:n_dyn]], slot[roots[n_dyn:n_dyn + n_char]], slot[roots[n_dyn + n_char:]]
    scene["model_lod"] = np.asarray([[0, 3], [1, 2], [0, 0]], np.uint8)
    scene["flags"] = (scene["flags"] & ~np.uint32(E_DIRTY)).astype(np.uint32)
    scene["flags"][scene["orig_of"] >= 0] |= np.uint32(E_DIRTY)           # first frame: everything is new
    cam = synth.camera(pos=(0, 20, 120))

    # ---- bodies: dynamic spheres drive dyn_e; character bodies are synced by the character feeder
    bodies = synth.sphere_bodies(n_dyn + n_char, box=30.0, seed=12)
    bodies["body_entity"] = np.concatenate([dyn_e, np.full(n_char, -1)]).astype(np.int32)
    feed = synth.character_feed(n_char, seed=12, with_bodies=True, body_base=n_dyn)
    feed["entity"] = char_e.astype(np.uint32)
    bodies["pos"][n_dyn:] = feed["pos"].astype(np.float64)
    bodies["pos"][n_dyn:, 1] += bodies["yoffset"][n_dyn:]
    link_body = (n_dyn + np.arange(n_char)).astype(np.uint32)           # characters' capsules follow the entity rotation
    link_entity = char_e.astype(np.uint32)

    # ---- lights carried by a few props
    L = synth.lights(24, seed=12, inactive_frac=0.1)
    carriers = dict(entity=prop_e[:6].astype(np.uint32), light=np.asarray([2, 3, 5, 7, 11, 3], np.int32),
                    off=np.random.Generator(np.random.PCG64(1)).uniform(-1, 1, (6, 3)).astype(np.float32))

    # ---- skinned characters
    sk = synth.skeleton(J, 5, seed=12)
    an = synth.animation(J, 8, 1.5, seed=12)
    ch = synth.characters(n_char, J, seed=12)
    sk["bind"] = ob.skeleton_bind(sk)
    mesh = synth.skinned_mesh(vpc, J, seed=12)
    vf, vc = np.zeros(n_char, np.uint32), np.full(n_char, vpc, np.uint32)

    # ---- particles
    ps = synth.particle_systems(n_sys=6, count=200, radius=3.0, velocity=0.5, seed=12)
    ppos, pvel, pst = ob.particles_spawn(ps, 0x1234ABCD330E)

    # ================= device side
    batch = entities.EntityBatch(scene, cuda_device)
    world = physics.PhysWorld(bodies, synth.static_boxes(8, 30.0), device=cuda_device)
    cf = characters.CharacterFeed(feed, cuda_device)
    ls = lights.LightSet(cuda_device, 1280, 720, 64)
    ls.load(L)
    ls.set_carriers(carriers["entity"], carriers["light"], carriers["off"])
    model = animation.SkinnedModel(sk, [an], mesh=mesh, bind=sk["bind"], device=cuda_device)
    cb = animation.CharacterBatch(model, n_char, ch["trs0"], batch.mx, entity_index=char_e, vert_first=vf, vert_count=vc)
    start = np.linspace(9.0, 10.0, n_char)
    cb.start_clock(ani_time=start, speed=np.full(n_char, 1.2, np.float32), repeat=np.ones(n_char, np.uint8))
    pb = particles.ParticleBatch(ps, ppos.copy(), pvel.copy(), pst, cuda_device)
    loop = frame.FrameLoop(batch, cam, world=world, feed=cf, body_links=(link_body, link_entity), lights=ls,
                           characters=cb, particles=pb, contacts=True, prebin=prebin)
    ref_world = physics.PhysWorld(bodies, synth.static_boxes(8, 30.0), device=cuda_device) if prebin else None

    # ================= oracle side (same order)
    o_scene = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
    st = ob.entity_state(o_scene)
    bst = ob.bodies_state(bodies)
    chars = dict(entity=feed["entity"], body=feed["body"], hist_pos=feed["hist_pos"].copy(),
                 hist_head=feed["hist_head"].copy(), hist_wrapped=feed["hist_wrapped"].copy(), airborne=feed["airborne"])
    o_lights = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in L.items()}
    time_acc = [0.0]
    ani = start.copy()
    trs = np.tile(ch["trs0"], (n_char, 1, 1))
    cur = np.zeros((n_char, J, 3), np.int32)
    cur_lod = np.zeros(o_scene["n"], np.int32)
    ofr, oview, oproj = ob.frustum_from_camera(cam)
    te = np.asarray([an["time_end"]], np.float32)

    now = 10.0
    for f in range(4):
        dt = [1 / 60, 1 / 120, 0.03, 1 / 60][f]
        now += dt
        loop.clap_frame(now, dt)

        steps, time_acc[0] = ob.phys_step_schedule(time_acc[0], dt)
        for _ in range(steps):
            ob.bodies_step(bodies, bst, 1.0 / 120.0)
        bview = dict(pos=bst["pos"], lvel=bst["lvel"], yoffset=np.ascontiguousarray(bodies["yoffset"]))
        ob.characters_update(chars, feed["limbo_height"], o_scene["pos_scale"], st["flags"], bview)
        ob.phys_body_update(bodies, bst, o_scene["pos_scale"], o_scene["rot"], st["flags"])
        dirty = ((st["flags"] & E_DIRTY) != 0).astype(np.uint8)
        ob.bodies_rotate_from_entities(link_body, link_entity, o_scene["rot"], o_scene["parent"], dirty, bst["quat"])
        o_lights["pos"] = ob.lights_from_entities(carriers, o_scene["pos_scale"], o_scene["parent"], dirty, o_lights)
        ob.entities_update(o_scene, st)
        ft, ended = ob.animation_time(np.zeros(n_char, np.uint32), te, ani, np.full(n_char, 1.2, np.float32),
                                      np.ones(n_char, np.uint8), now)
        cur[ended != 0] = 0
        jt, _gl, jp = ob.pose(sk, an, ft, st["mx"][char_e], trs, cur)
        sp, sn = ob.skin(mesh, vf, vc, jt)
        _k, pst = ob.particles_update(ps, ppos, pvel, pst)
        tiles = ob.light_grid_compute(o_lights, oview, oproj, 1280, 720, 64)
        vis, _m = ob.entities_cull(o_scene["n"], st["flags"], st["aabb"], ofr)
        draw = ob.entities_lod(o_scene, st, vis, cam["cam_pos"], scene["model_lod"], np.full(o_scene["n"], -1, np.int32), cur_lod)

        out = batch.download()
        assert np.array_equal(out["mx"].view(np.uint32), st["mx"].view(np.uint32)), f"frame {f} mx"
        assert np.array_equal(out["inv_mx"].view(np.uint32), st["inv_mx"].view(np.uint32)), f"frame {f} inverse"
        assert np.array_equal(out["aabb"].view(np.uint32), st["aabb"].view(np.uint32)), f"frame {f} aabb"
        assert np.array_equal(out["visible"], vis), f"frame {f} visible list"
        assert np.array_equal(batch.draw_lod[:len(vis)].cpu().numpy(), draw), f"frame {f} draw LODs"
        w = world.download()
        assert np.array_equal(w["pos"].view(np.uint64), bst["pos"].view(np.uint64)), f"frame {f} body positions"
        assert np.array_equal(w["quat"].view(np.uint64), bst["quat"].view(np.uint64)), f"frame {f} body quaternions"
        c = cf.download()
        assert np.array_equal(c["hist_head"], chars["hist_head"]) and np.array_equal(c["hist_pos"], chars["hist_pos"])
        assert np.array_equal(ls.download_pos()[:24], o_lights["pos"]), f"frame {f} light positions"
        assert np.array_equal(ls.download_tiles(), tiles), f"frame {f} light grid"
        cd = cb.download()
        reach = sk["order"]
        assert bits_equal(cd["joint_transforms"][:, reach], jt[:, reach]), f"frame {f} palette"          # bit-exact (round 4)
        assert bits_equal(cd["out_position"], sp), f"frame {f} skinned verts"
        assert np.array_equal(cb.download_clock()["ani_time"], ani)
        p = pb.download()
        assert np.array_equal(p["pos"].view(np.uint32), ppos.view(np.uint32)) and p["rng_state"] == pst, f"frame {f} particles"
        # game code between frames: push a few props around (entity3d_position) -> dirty
        if f < 3:
            idx = prop_e[f::5][:10]
            newp = o_scene["pos_scale"][idx].copy()
            newp[:, 0] += np.float32(1.5)
            o_scene["pos_scale"][idx] = newp
            st["flags"][idx] |= np.uint32(E_DIRTY)
            batch.set_transforms(idx, newp, o_scene["rot"][idx])
    assert world.download()["pair_total"] >= 0
    if prebin:                                               # the last frame's pairs against a plain broadphase over the same boxes
        wd = world.download()
        ref_world.aabb[:world.n].copy_(world.aabb[:world.n])
        # the frame's collide ran BEFORE its step: redo it on both over the final boxes (the frame's step pre-binned them)
        world.broadphase(); ref_world.broadphase()
        a, b = world.download(), ref_world.download()
        assert a["pair_total"] == b["pair_total"] and np.array_equal(a["pairs"], b["pairs"])
        assert np.array_equal(a["static_pairs"], b["static_pairs"])


def test_captured_frame_graph_replays_the_same_frames(cuda_device):
    """FrameLoop.capture(): one frame recorded as a HIP graph (device-resident clock) gives, replayed,
    exactly what issuing the launches one by one gives."""
    import torch
    from clap_amd import animation, characters, entities, frame, lights, particles, physics, tiler

    def build():
        raw = synth.entities_flat(3000, seed=5)
        scene, tl = tiler.tiled_scene(raw)
        roots = tl["slot_of"][np.flatnonzero(raw["parent"] < 0)]
        scene["model_lod"] = np.asarray([[0, 3]], np.uint8)
        batch = entities.EntityBatch(scene, cuda_device)
        nb, nc, J, vpc = 200, 12, 24, 90
        b = synth.sphere_bodies(nb, box=10.0, seed=5)
        b["body_entity"] = roots[:nb].astype(np.int32)
        world = physics.PhysWorld(b, synth.static_boxes(6, 10.0), pair_capacity=8192, device=cuda_device)
        feed = synth.character_feed(nc, seed=5, with_bodies=False)
        feed["entity"] = roots[nb:nb + nc].astype(np.uint32)
        cf = characters.CharacterFeed(feed, cuda_device)
        ls = lights.LightSet(cuda_device, 1280, 720, 64)
        ls.load(synth.lights(12, seed=5))
        ls.set_carriers(roots[-4:].astype(np.uint32), np.arange(4, dtype=np.int32), np.ones((4, 3), np.float32))
        sk, an = synth.skeleton(J, 5, seed=5), synth.animation(J, 8, 0.05, seed=5)
        ch = synth.characters(nc, J, seed=5)
        model = animation.SkinnedModel(sk, [an], mesh=synth.skinned_mesh(vpc, J, seed=5), device=cuda_device)
        cb = animation.CharacterBatch(model, nc, ch["trs0"], batch.mx, entity_index=feed["entity"],
                                      vert_first=np.zeros(nc, np.uint32), vert_count=np.full(nc, vpc, np.uint32))
        cb.start_clock(ani_time=np.zeros(nc), speed=np.ones(nc, np.float32))
        ps = synth.particle_systems(n_sys=3, count=256, radius=2.0, velocity=0.5, seed=5)
        ppos, pvel, pst = ob.particles_spawn(ps, 0x1234ABCD330E)
        pb = particles.ParticleBatch(ps, ppos, pvel, pst, cuda_device)
        loop = frame.FrameLoop(batch, synth.camera(pos=(0, 10, 60)), world=world, feed=cf, lights=ls, characters=cb,
                               particles=pb, contacts=True)
        return loop

    def state(loop):
        torch.cuda.synchronize()
        e, c, p, w = loop.batch.download(), loop.characters.download(), loop.particles.download(), loop.world.download()
        return dict(mx=e["mx"], visible=e["visible"], lod=loop.batch.draw_lod[:len(e["visible"])].cpu().numpy(),
                    jt=c["joint_transforms"], jpos=c["joint_pos"], skinned=c["out_position"], ani=loop.characters.download_clock()["ani_time"],
                    ppos=p["pos"], rng=np.asarray([p["rng_state"]], np.uint64), bpos=w["pos"], pairs=w["pairs"],
                    spairs=w["static_pairs"], tiles=loop.lights.download_tiles())

    # eager / graph: CLAPGPU_FRAME_OVERLAP -- three chains on the caller's stream and two helper streams, joined by events
    # (frame.hip) -- issued and replayed; single: the default, every launch on one stream in the reference's order
    eager, graph, single = build(), build(), build()
    eager.overlap = graph.overlap = True
    dt = 1.0 / 120.0
    eager.clap_frame(dt, dt)                                # frame 1 on all (capture() issues its warm-up frame eagerly)
    single.clap_frame(dt, dt)
    graph.capture(dt, warmup_now=dt)
    for f in range(2, 12):
        eager.clap_frame(f * dt, dt)
        single.clap_frame(f * dt, dt)
        graph.clap_frame_replay(f * dt)
        a, b, c = state(eager), state(graph), state(single)
        for k in a:
            assert np.array_equal(a[k], b[k]), f"frame {f}: {k} (graph replay)"
            assert np.array_equal(a[k], c[k]), f"frame {f}: {k} (three chains vs one stream)"
    single.capture(dt, warmup_now=12 * dt)                  # the one-stream frame captures into a graph too
    eager.clap_frame(12 * dt, dt)
    for f in range(13, 16):
        eager.clap_frame(f * dt, dt)
        single.clap_frame_replay(f * dt)
        a, c = state(eager), state(single)
        for k in a:
            assert np.array_equal(a[k], c[k]), f"frame {f}: {k} (one-stream graph replay)"
    assert (state(graph)["ani"] != 0).any(), "the 0.05 s animation restarted during the replayed frames"
