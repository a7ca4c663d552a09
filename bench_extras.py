#!/usr/bin/env python3
"""bench_extras.py -- the secondary measurements of bench.py's line (everything that is not the headline's timed loop):
per-kernel rows of `extra` (pose, skinning, particles, bodies / broadphase / contacts, light grid), the whole frame at
BASELINE sizes and at testbed size, and the two digests of them, `summary` and `roofline.secondary`.  GPU side only: the
CPU legs (`cpu_baseline`, `dropin_boundary`: the only users of oracle/) stay in bench.py.  Imported by bench.py and by
tools/run_kernel.py; not a program.
"""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def time_launches(fn, iters, warmup=10):
    """Mean duration (s) of fn() measured with HIP events on torch's current stream."""
    import torch
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3


_SPAWNED = {}


def spawn_cached(ps, state):
    """synth.particles_spawn (the harness's own host-side spawn), once per particle-system table."""
    from clap_amd import synth
    key = (ps["sys"].tobytes(), int(state))
    if key not in _SPAWNED:
        _SPAWNED.clear()
        _SPAWNED[key] = synth.particles_spawn(ps, state)
    pos, vel, st = _SPAWNED[key]
    return pos.copy(), vel.copy(), st


def pmc_kernel_traffic(*names):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes) of the named kernels from the
    newest committed summary profiles/<round>_*/pmc_hbm_bytes.json (tools/profile_round.sh), or None."""
    import glob
    best = None
    # by name: r02_a < r02_b < ... < r03_a; the newest summary wins (an older round's stands until this round has one)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*", "pmc_hbm_bytes.json"))):
        try:
            with open(path) as f:
                best = json.load(f).get("kernels", {})
            # the headline kernel from the headline-only passes of the same round (the whole-frame extra launches the
            # same grid with part of the scene dirty)
            hp = os.path.join(os.path.dirname(path), "entities_pmc.json")
            if os.path.exists(hp):
                with open(hp) as f:
                    h = json.load(f)
                for k in list(best):
                    if h.get("kernel", "\0") in k:
                        best[k] = dict(best[k], hbm_bytes_per_launch=h["hbm_bytes_per_launch"])
        except (OSError, ValueError):
            pass
    if not best:
        return None
    total = 0.0
    for want in names:
        hit = [v for k, v in best.items() if want in k]
        if not hit:
            return None
        total += hit[0]["hbm_bytes_per_launch"]
    return total


def roof(alg_bytes, seconds, *kernels):
    gbs = alg_bytes / seconds / 1e9
    traffic = pmc_kernel_traffic(*kernels) if kernels else None
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            # bytes that really moved (PMC, profiles/<round>_*/pmc_hbm_bytes.json of this tree) / this run's launch time / peak
            "moved_frac": None if traffic is None else traffic / seconds / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_us": seconds * 1e6}


def snapshot_characters(comps, raw, device, steps, warmup):
    """--snapshot: pose + skinning of the snapshot's own skinned models (a scene loaded from scene.json + glTF by
    clapgpu_load_scene carries them as model<k>.* and characters.*), one batch per model, timed like a step."""
    import torch
    from clap_amd import animation, snapshot
    models = snapshot.skinned_models(comps)
    chars = comps.get("characters")
    if not models or chars is None or not len(chars["entity"]):
        return None
    out = []
    for k, (sk, anims, mesh) in models.items():
        ents = np.asarray(chars["entity"])[np.asarray(chars["model"]) == k]
        if not len(ents) or not anims:
            continue
        n, J, V = len(ents), sk["nr_joints"], mesh["n_verts"]
        mx = np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1))
        mx[:, 12:15] = raw["pos_scale"][ents, :3]
        model = animation.SkinnedModel(sk, anims, mesh=mesh, bind=sk["bind"], device=device)
        cb = animation.CharacterBatch(model, n, np.zeros((J, 10), np.float32), mx, vert_first=np.zeros(n, np.uint32),
                                      vert_count=np.full(n, V, np.uint32))
        cb.set_frame_times(np.linspace(0.0, float(anims[0]["time_end"]), n, dtype=np.float32))

        def step():
            cb.pose_update()
            cb.skin()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out.append({"model": int(k), "characters": n, "joints": J, "vertices_per_character": V, "animations": len(anims),
                    "us_per_step": dt * 1e6, "skinned_verts_per_s": n * V / dt})
    return out


def extras(device, testbed=True):
    """The other rows of the hot path at BASELINE configs[2] and configs[3] sizes, each as
    units/s plus the algorithmic-bytes roofline of its kernel (SURVEY.md 8d byte counts)."""
    import torch
    from clap_amd import animation, particles, physics, synth
    out = {}

    # ---- configs[2]: 50k characters x 64 joints, 10M vertices (one 200-vertex mesh per character) ----
    J, n_chars, vpc = 64, 50_000, 200
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n_chars, J, seed=3)
    mesh = synth.skinned_mesh(vpc, J, seed=3, copies=n_chars)          # distinct vertices per character
    vf = (np.arange(n_chars, dtype=np.int64) * vpc).astype(np.uint32)
    vc = np.full(n_chars, vpc, np.uint32)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, device=device)
    cb = animation.CharacterBatch(model, n_chars, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
    cb.set_frame_times(ch["phase"])
    # these two settle only after some tens of launches of a process (the first ones run 10-20 us slower): steady state
    t_pose = time_launches(cb.pose_update, 200, warmup=100)
    t_skin = time_launches(cb.skin, 200, warmup=100)
    out["pose_palette"] = {"joints_per_s": n_chars * J / t_pose, "characters": n_chars, "joints": J,
                           "kernel": "k_pose<64, 768, false, true> (the reference's arithmetic: results bit-exact, signed zeros included)",
                           "launches_timed": 200,
                           "roofline": roof(cb.pose_algorithmic_bytes(), t_pose, "[all outputs]"),
                           "note": "frac prices SURVEY 8d's 200 B/joint, of which 80 B are keyframes that are per MODEL and "
                                   "come from LDS / L2: moved_frac (PMC bytes of the ALL-OUTPUTS launches / this run's time / "
                                   "8 TB/s) is the HBM utilisation"}
    # what a frame whose skinning runs on the device needs from the pose: the palette alone.  The joints' T/R/S and
    # world positions (56 B of the 120 B a joint writes) are host-visible state of animated_update; a caller that does
    # not read them back switches them off (clapgpu_pose_batch.skip)
    cb.set_outputs(trs=False, joint_pos=False)
    t_pal = time_launches(cb.pose_update, 200, warmup=100)
    cb.set_outputs(trs=True, joint_pos=True)
    pal_traffic = pmc_kernel_traffic("[palette only]")
    out["pose_palette"]["palette_only"] = {"us": t_pal * 1e6, "joints_per_s": n_chars * J / t_pal,
                                           "bytes_written_per_joint": 64, "traffic": pal_traffic,
                                           "moved_frac": None if pal_traffic is None else pal_traffic / t_pal / 1e9 / HBM_PEAK_GBS,
                                           "note": "CLAPGPU_POSE_SKIP_TRS | CLAPGPU_POSE_SKIP_JOINT_POS"}
    out["skinning"] = {"skinned_verts_per_s": n_chars * vpc / t_skin, "vertices": n_chars * vpc,
                       "kernel": "k_skin", "roofline": roof(cb.skin_algorithmic_bytes(), t_skin, "k_skin"),
                       "mesh": "one distinct 200-vertex mesh per character (44 B/vertex read from HBM); instanced "
                               "meshes read less"}
    out["pose_plus_skinning_verts_per_s"] = n_chars * vpc / (t_pose + t_skin)
    del cb, model, mesh
    torch.cuda.empty_cache()

    # ---- configs[3], particle half: 4096 systems x 1024 particles ----
    ps = synth.particle_systems(n_sys=4096, count=1024, radius=10.0, velocity=0.005, dist=synth.PART_DIST_SQRT)
    pos, vel, st = spawn_cached(ps, synth.DRAND48_DEFAULT_STATE)
    pb = particles.ParticleBatch(ps, pos, vel, st, device)
    view = np.eye(4, dtype=np.float32).ravel()
    t_part = time_launches(lambda: pb.particles_update(view), 200, warmup=100)
    out["particles"] = {"particles_per_s": pb.n_real / t_part, "particles": pb.n_real,
                        "kernels": "k_particles_advect + k_visible_expand_rp + k_particles_respawn",
                        "roofline": roof(pb.algorithmic_bytes(), t_part, "k_particles_advect", "k_particles_respawn_rp")}
    del pb
    # ---- configs[3], body half: 256k bodies with the reference's geoms (capsules, "puppy" capsules, spheres): integrate +
    #      both broadphase passes + narrowphase ----
    b = synth.capsule_bodies(262_144, box=60.0, seed=4)
    pw = physics.PhysWorld(b, synth.static_boxes(64, 60.0), pair_capacity=2_000_000, device=device)
    t_int = time_launches(lambda: pw.world_step(1.0 / 120.0), 200, warmup=100)
    t_bp = time_launches(pw.broadphase, 100, warmup=50)
    npairs = int(pw.pair_total.item())
    t_con = time_launches(pw.contacts_geoms_both, 100, warmup=30)            # near_callback over both candidate lists
    out["bodies"] = {"bodies_per_s_integrate": pw.n / t_int, "bodies": pw.n, "kernel": "k_bodies_step",
                     "roofline": roof(pw.integrate_algorithmic_bytes(), t_int, "k_bodies_step"),
                     "note": "the step also keeps the narrowphase's 64-byte geom record of every body (clapgpu_bodies.geom_records: "
                             "+64 B / body written, outside SURVEY's 232 B): 18.5 -> 22.9 us here, 50 -> 39 us in the contact kernel",
                     "contacts": {"us": t_con * 1e6, "kernel": "k_contacts_geoms_both", "candidate_pairs": npairs,
                                  "static_candidate_pairs": int(pw.static_pair_total.item()),
                                  # per candidate pair: its 8 bytes, two 64-byte geoms read, one 160-byte record written
                                  "roofline": roof((npairs + int(pw.static_pair_total.item())) * (8 + 2 * 64 + 160), t_con, "k_contacts_geoms_both"),
                                  "note": "both candidate lists of a substep in one launch, 160-byte records; a body geom is read as "
                                          "one 64-byte record"},
                     "broadphase": {"bodies_per_s": pw.n / t_bp, "pairs": npairs, "ms": t_bp * 1e3,
                                    "algorithmic_bytes": 24 * pw.n + 8 * npairs,
                                    "launches": 5,
                                    "moved_frac": (lambda t: None if t is None else t / t_bp / 1e9 / HBM_PEAK_GBS)(
                                        pmc_kernel_traffic("k_bp_bin", "k_bp_cells", "k_bp_scatter", "k_bp_search", "k_bp_emit")),
                                    "note": "both passes (bodies x bodies, statics x bodies) in the same five launches: "
                                            "k_bp_bin, k_bp_cells, k_bp_scatter, k_bp_search, k_bp_emit; bound "
                                            "by the fabric's atomic rate (bin), launch floors and the search's chain of "
                                            "dependent steps, not by HBM (profiles/r02_experiments/broadphase_tiles.md)"}}
    del pw
    # ---- 8f rank 2: clustered-lighting tile masks, 128 light slots x a 4K screen at the reference's 64-px tiles ----
    from clap_amd import lights as gl
    ls = gl.LightSet(device, 3840, 2160, gl.TILE_WIDTH)
    ls.load(synth.lights(128, seed=7))
    from clap_amd import entities as _ent
    _fr, vm, pm = _ent.view_calc_frustum(synth.camera(pos=(1.0, 2.0, 3.0)))
    t_lg = time_launches(lambda: ls.grid_compute(vm, pm), 30)
    tw, th = gl.grid_dims(3840, 2160, gl.TILE_WIDTH)
    out["light_grid"] = {"tiles_per_s": tw * th / t_lg, "tiles": tw * th, "lights": 128, "us": t_lg * 1e6,
                         "kernel": "k_light_grid", "note": "launch-bound: 2040 tiles x 128 lights, 32 KB out"}
    del ls
    torch.cuda.empty_cache()
    out["full_frame"] = full_frame(device)
    if testbed:
        out["testbed_frame"] = testbed_frame(device)
    return out


def testbed_frame(device):
    """BASELINE configs[0] scale (the reference's own CPU-runnable case): 10k flat entities alone, and a whole
    testbed-sized frame (10 characters, 128 bodies, 8 particle systems, 16 lights) issued launch by launch and
    replayed as a captured HIP graph.  This regime is launch latency, not bytes."""
    import torch
    from clap_amd import animation, characters, entities, frame, lights, particles, physics, synth, tiler
    raw = synth.entities_flat(10_000, seed=1234)
    scene, tl = tiler.tiled_scene(raw)
    roots = tl["slot_of"][np.flatnonzero(raw["parent"] < 0)]
    batch = entities.EntityBatch(scene, device)
    cam = synth.camera()
    fr, _v, _p = entities.view_calc_frustum(cam)

    def ent_step():
        batch.mq_update(fr, all_dirty=True)
        batch.compact_visible()
    t_ent = time_launches(ent_step, 200, warmup=50)
    n_bodies, n_chars, J, vpc = 128, 10, 64, 2000
    b = synth.sphere_bodies(n_bodies, box=16.0, seed=4)
    b["body_entity"] = roots[:n_bodies].astype(np.int32)
    world = physics.PhysWorld(b, synth.static_boxes(16, 16.0), pair_capacity=4096, device=device)
    feed = synth.character_feed(n_chars, seed=13, with_bodies=False)
    feed["entity"] = roots[n_bodies:n_bodies + n_chars].astype(np.uint32)
    cf = characters.CharacterFeed(feed, device)
    ls = lights.LightSet(device, 1920, 1080, lights.TILE_WIDTH)
    ls.load(synth.lights(16, seed=7))
    ls.set_carriers(roots[-8:].astype(np.uint32), np.arange(8, dtype=np.int32), np.zeros((8, 3), np.float32))
    sk, an = synth.skeleton(J, 8, seed=3), synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n_chars, J, seed=3)
    model = animation.SkinnedModel(sk, [an], mesh=synth.skinned_mesh(vpc, J, seed=3), device=device)
    cb = animation.CharacterBatch(model, n_chars, ch["trs0"], batch.mx, entity_index=feed["entity"],
                                  vert_first=np.zeros(n_chars, np.uint32), vert_count=np.full(n_chars, vpc, np.uint32))
    cb.start_clock(ani_time=-ch["phase"].astype(np.float64), speed=np.ones(n_chars, np.float32))
    ps = synth.particle_systems(n_sys=8, count=1024, radius=10.0, velocity=0.005)
    pos, vel, st = spawn_cached(ps, synth.DRAND48_DEFAULT_STATE)
    pb = particles.ParticleBatch(ps, pos, vel, st, device)
    loop = frame.FrameLoop(batch, cam, world=world, feed=cf, lights=ls, characters=cb, particles=pb, contacts=True)
    now = [0.0]

    def one():
        now[0] += 1.0 / 120.0
        loop.clap_frame(now[0], 1.0 / 120.0)
    t_frame = time_launches(one, 100, warmup=50)
    t_graph = None
    try:
        loop.capture(1.0 / 120.0, warmup_now=now[0] + 1.0 / 120.0)
        now[0] += 1.0 / 120.0

        def replay():
            now[0] += 1.0 / 120.0
            loop.clap_frame_replay(now[0])
        t_graph = time_launches(replay, 100, warmup=20)
    except Exception as exc:
        print(f"[bench] testbed frame graph capture failed: {exc}", file=sys.stderr)
    return {"entities": 10_000, "entity_step_us": t_ent * 1e6, "entity_updates_per_s": 10_000 / t_ent,
            "frame_us": t_frame * 1e6, "frame_graph_replay_us": None if t_graph is None else t_graph * 1e6,
            "contents": "10k flat entities + 10 characters x 64 joints / 20k skinned vertices + 128 bodies + 8k particles + "
                        "16 lights; launch-latency bound (the reference's CPU path needs ~1.1 ms for the 10k entities alone)"}


def full_frame(device):
    """One clap_frame() of everything at BASELINE sizes at once (clap_amd.frame.FrameLoop): configs[1]'s 1M-entity
    hierarchy, configs[2]'s 50k characters (feeder, clock, pose, 10M skinned vertices), configs[3]'s 256k bodies
    (broadphase x2, contacts, integrate, read-back into 75k entities) and 4M particles, 128 lights."""
    import torch
    from clap_amd import animation, characters, entities, frame, lights, particles, physics, synth, tiler
    raw = synth.entities_chains(125_000, 8, seed=2)
    scene, tl = tiler.tiled_scene(raw)
    roots = tl["slot_of"][np.flatnonzero(raw["parent"] < 0)]
    batch = entities.EntityBatch(scene, device)
    cam = synth.camera()
    n_bodies, n_bound, n_chars, J, vpc = 262_144, 75_000, 50_000, 64, 200
    b = synth.capsule_bodies(n_bodies, box=60.0, seed=4)
    b["body_entity"] = np.concatenate([roots[:n_bound], np.full(n_bodies - n_bound, -1)]).astype(np.int32)
    world = physics.PhysWorld(b, synth.static_boxes(64, 60.0), pair_capacity=2_000_000, device=device)
    feed = synth.character_feed(n_chars, seed=13, with_bodies=False)
    feed["entity"] = roots[n_bound:n_bound + n_chars].astype(np.uint32)
    cf = characters.CharacterFeed(feed, device)
    ls = lights.LightSet(device, 3840, 2160, lights.TILE_WIDTH)
    ls.load(synth.lights(128, seed=7))
    ls.set_carriers(roots[-64:].astype(np.uint32), np.arange(64, dtype=np.int32), np.zeros((64, 3), np.float32))
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n_chars, J, seed=3)
    mesh = synth.skinned_mesh(vpc, J, seed=3, copies=n_chars)
    vf = (np.arange(n_chars, dtype=np.int64) * vpc).astype(np.uint32)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, device=device)
    cb = animation.CharacterBatch(model, n_chars, ch["trs0"], batch.mx, entity_index=feed["entity"], vert_first=vf,
                                  vert_count=np.full(n_chars, vpc, np.uint32))
    cb.start_clock(ani_time=-ch["phase"].astype(np.float64), speed=np.ones(n_chars, np.float32))
    ps = synth.particle_systems(n_sys=4096, count=1024, radius=10.0, velocity=0.005, dist=synth.PART_DIST_SQRT)
    pos, vel, st = spawn_cached(ps, synth.DRAND48_DEFAULT_STATE)
    pb = particles.ParticleBatch(ps, pos, vel, st, device)
    # no reader of the joints' T / R / S or positions inside the frame: the skinning takes the palette (model.c:1020-1022)
    # ... and nothing but the frame writes the bodies' boxes: the step bins them for the next broadphase (CLAPGPU_FRAME_PREBIN)
    loop = frame.FrameLoop(batch, cam, world=world, feed=cf, lights=ls, characters=cb, particles=pb, contacts=True, pose_readers=(),
                           prebin=True)
    now = [0.0]

    def one():
        now[0] += 1.0 / 120.0
        loop.clap_frame(now[0], 1.0 / 120.0)                 # one physics substep per frame
    t = time_launches(one, 40, warmup=40)                   # the first frames of a process run several times slower
    cb.set_outputs(trs=True, joint_pos=True)                # ... and with every by-product of the pose written (round 4's frame)
    t_all_outputs = time_launches(one, 40, warmup=10)
    cb.set_outputs(trs=False, joint_pos=False)
    if os.environ.get("CLAP_FRAME_ONE_STREAM_ONLY") == "1":  # a kernel trace of the one-stream frame alone (tools/r05/prof_frame.sh)
        return {"ms_per_frame": t * 1e3}
    loop.overlap = True                                     # the same frame as three chains on three streams (frame.hip)
    t_overlap = time_launches(one, 40, warmup=10)
    loop.overlap = False
    graph_ms = None
    try:                                                     # the same frame as one captured HIP graph
        loop.capture(1.0 / 120.0, warmup_now=now[0] + 1.0 / 120.0)
        now[0] += 1.0 / 120.0

        def replay():
            now[0] += 1.0 / 120.0
            loop.clap_frame_replay(now[0])
        graph_ms = time_launches(replay, 40, warmup=10) * 1e3
    except Exception as exc:                                 # informational leg: never fail the benchmark on it
        print(f"[bench] frame graph capture failed: {exc}", file=sys.stderr)
    return {"ms_per_frame": t * 1e3, "frames_per_s": 1.0 / t, "ms_per_frame_three_streams": t_overlap * 1e3,
            "ms_per_frame_graph_replay": graph_ms, "ms_per_frame_all_pose_outputs": t_all_outputs * 1e3,
            "pose": "palette only (CLAPGPU_POSE_SKIP_TRS | CLAPGPU_POSE_SKIP_JOINT_POS): nothing in the frame reads the joints' "
                    "T / R / S or world positions; ms_per_frame_all_pose_outputs = with them written (round 4's frame)",
            "label": "physics WITHOUT contact response: not a whole clap_frame()",
            "contents": "1M entities (depth 8) + 50k characters x 64 joints + 10M skinned vertices + 262144 bodies "
                        "(75k bound to entities; 2 broadphase passes, contact generation, integrate) + 4M particles + 128 "
                        "lights on a 4K light grid; one physics substep per frame.  The bodies are integrated as "
                        "constraint-free: the contact pairs the same frame generates are handed to nobody -- the SOR-LCP "
                        "that would apply them is ODE's dWorldQuickStep (physics.c:769), outside this path's scope",
            "launches": "one stream, no host read-back inside the frame.  ms_per_frame_three_streams: the same frame as three "
                        "chains (physics -> entities | clock -> pose -> skinning | particles) on the caller's stream and two "
                        "helper streams (CLAPGPU_FRAME_OVERLAP): they overlap and the frame is no shorter (csrc/frame.hip)"}


def summary(out, extra):
    """<= 15 scalars: what the long `extra` / `cpu_baseline` sections say, where a record that keeps only the head of the
    line (or drops `extra`) still has them.  Microseconds per launch unless the key says otherwise."""
    s = {}

    def put(key, fn):
        try:
            v = fn()
            if v is not None:
                s[key] = round(float(v), 3)
        except (KeyError, TypeError, IndexError):
            pass
    e = extra or {}
    put("pose_us", lambda: e["pose_palette"]["characters"] * e["pose_palette"]["joints"] / e["pose_palette"]["joints_per_s"] * 1e6)
    put("pose_palette_us", lambda: e["pose_palette"]["palette_only"]["us"])
    put("skin_us", lambda: e["skinning"]["vertices"] / e["skinning"]["skinned_verts_per_s"] * 1e6)
    put("particles_us", lambda: e["particles"]["particles"] / e["particles"]["particles_per_s"] * 1e6)
    put("step_us", lambda: e["bodies"]["bodies"] / e["bodies"]["bodies_per_s_integrate"] * 1e6)
    put("bp_us", lambda: e["bodies"]["broadphase"]["ms"] * 1e3)
    put("contacts_us", lambda: e["bodies"]["contacts"]["us"])
    put("frame_ms", lambda: e["full_frame"]["ms_per_frame"])
    b = ((out.get("cpu_baseline") or {}).get("dropin_boundary") or {}).get("1000000_entities_100pct_dirty") or {}
    d = b.get("scatter_drawn") or {}
    put("boundary_1m_frame_ms", lambda: d["binding_frame_draw_list_ms"])
    put("boundary_1m_mq_update_ms", lambda: d["binding_mq_update_ms"])
    put("boundary_1m_frame_ms_scatter_all", lambda: b["binding_frame_draw_list_ms"])
    put("boundary_1m_reference_frame_ms", lambda: b["reference_frame_ms"])
    c = ((out.get("cpu_baseline") or {}).get("dropin_boundary") or {}).get("1000000_entities_10pct_dirty_10_made_10_deleted_a_frame") or {}
    put("boundary_1m_churn_mq_update_ms", lambda: c["binding_mq_update_ms"])
    put("boundary_1m_churn_reference_mq_update_ms", lambda: c["reference_mq_update_ms"])
    w = ((out.get("cpu_baseline") or {}).get("dropin_boundary") or {}).get("1000000_entities_10pct_dirty_no_notifications") or {}
    put("boundary_1m_no_notify_mq_update_ms", lambda: w["binding_mq_update_ms"])
    return s


def secondary(out, extra):
    """The secondary rows where the driver's record keeps them whole: inside `roofline` (its `parsed` drops `summary` and
    `extra`).  Per kernel group {us, frac, moved_frac} -- microseconds per launch group, algorithmic bytes / time / 8 TB/s,
    PMC bytes of the committed profile / this run's time / 8 TB/s -- the frame, and the boundary at a million entities."""
    e = extra or {}
    sec = {}

    def grp(key, us, roofline):
        try:
            u = us()
            r = roofline() or {}
            sec[key] = {"us": round(float(u), 2), "frac": None if r.get("frac") is None else round(float(r["frac"]), 4),
                        "moved_frac": None if r.get("moved_frac") is None else round(float(r["moved_frac"]), 4)}
        except (KeyError, TypeError, IndexError, ZeroDivisionError):
            pass
    grp("pose", lambda: e["pose_palette"]["characters"] * e["pose_palette"]["joints"] / e["pose_palette"]["joints_per_s"] * 1e6,
        lambda: e["pose_palette"]["roofline"])
    grp("pose_palette", lambda: e["pose_palette"]["palette_only"]["us"],
        lambda: {"frac": 64 * e["pose_palette"]["characters"] * e["pose_palette"]["joints"] / (e["pose_palette"]["palette_only"]["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "moved_frac": e["pose_palette"]["palette_only"]["moved_frac"]})
    grp("skin", lambda: e["skinning"]["vertices"] / e["skinning"]["skinned_verts_per_s"] * 1e6, lambda: e["skinning"]["roofline"])
    grp("particles", lambda: e["particles"]["particles"] / e["particles"]["particles_per_s"] * 1e6, lambda: e["particles"]["roofline"])
    grp("step", lambda: e["bodies"]["bodies"] / e["bodies"]["bodies_per_s_integrate"] * 1e6, lambda: e["bodies"]["roofline"])
    grp("bp", lambda: e["bodies"]["broadphase"]["ms"] * 1e3,
        lambda: {"frac": e["bodies"]["broadphase"]["algorithmic_bytes"] / (e["bodies"]["broadphase"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "moved_frac": e["bodies"]["broadphase"].get("moved_frac")})
    grp("contacts", lambda: e["bodies"]["contacts"]["us"], lambda: e["bodies"]["contacts"].get("roofline"))
    try:
        sec["frame_ms"] = round(float(e["full_frame"]["ms_per_frame"]), 4)
    except (KeyError, TypeError):
        pass
    db = (out.get("cpu_baseline") or {}).get("dropin_boundary") or {}

    def row(src, *keys):
        d = {}
        for name, path in keys:
            try:
                v = src
                for k in path:
                    v = v[k]
                d[name] = round(float(v), 3)
            except (KeyError, TypeError):
                pass
        return d
    b = db.get("1000000_entities_100pct_dirty") or {}
    c = db.get("1000000_entities_10pct_dirty_10_made_10_deleted_a_frame") or {}
    w = db.get("1000000_entities_10pct_dirty_no_notifications") or {}
    m = row(b, ("frame_ms", ("scatter_drawn", "binding_frame_draw_list_ms")), ("mq_update_ms", ("scatter_drawn", "binding_mq_update_ms")),
            ("reference_frame_ms", ("reference_frame_ms",)), ("frame_ms_scatter_all", ("binding_frame_draw_list_ms",)))
    m.update(row(c, ("churn_mq_update_ms", ("binding_mq_update_ms",)), ("churn_reference_mq_update_ms", ("reference_mq_update_ms",))))
    m.update(row(w, ("no_notify_mq_update_ms", ("binding_mq_update_ms",)), ("no_notify_reference_mq_update_ms", ("reference_mq_update_ms",))))
    # the frames that walk the queue: walked + re-tiled every frame, and walked with the layout standing
    rt = db.get("1000000_entities_10pct_dirty_walked_and_retiled_every_frame") or {}
    wk = db.get("1000000_entities_10pct_dirty_walked_every_frame") or {}
    m.update(row(rt, ("walked_retiled_mq_update_ms", ("binding_mq_update_ms",)), ("walked_retiled_walk_ms", ("binding_ms", "walk")),
                 ("walked_retiled_reference_mq_update_ms", ("reference_mq_update_ms",))))
    m.update(row(wk, ("walked_mq_update_ms", ("binding_mq_update_ms",)), ("walked_walk_ms", ("binding_ms", "walk")),
                 ("walked_reference_mq_update_ms", ("reference_mq_update_ms",))))
    if m:
        sec["boundary_1m"] = m
    for n, name in ((1_000_000, "pipeline_frame_1m"), (10_000, "pipeline_frame_10k")):
        pf = db.get(f"{n}_entities_10pct_dirty_pipeline_frame_2_shadow_passes") or {}
        r = row(pf, ("frame_ms", ("binding_pipeline_frame_draw_list_ms",)), ("frame_block_ms", ("binding_pipeline_frame_block_ms",)),
                ("reference_frame_ms", ("reference_pipeline_frame_ms",)), ("mq_update_ms", ("binding_mq_update_ms",)),
                ("views_culled_per_update", ("views_culled_per_update",)), ("cull_launches_after_update", ("cull_launches_after_update",)))
        if r:
            r["identical"] = bool(pf.get("identical"))
            sec[name] = r
    return sec
