#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the REAL reference code.

Runs oracle/_ref/clap_ref (the reference's own core/model.c, view.c, transform.c,
particle.c compiled from /root/reference by oracle/ref/Makefile) on small seeded
inputs and stores inputs + the reference's outputs as data fixtures.  Only usable
in the build container (needs /root/reference); the fixtures are what travels.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from clap_amd import synth  # noqa: E402
from oracle import refrun  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

SCENE_KEYS = ("pos_scale", "rot", "parent", "model", "model_aabb", "model_skip", "flags", "level_start")
CAM_KEYS = ("cam_pos", "cam_quat", "persp", "ndc_z_zero_one")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{path}: {os.path.getsize(path)} bytes")


def entity_fixture(name, scene, cam, frames=None):
    scene = synth.pad_levels(scene)
    ref = refrun.entities(scene, cam, frames)
    d = {"in_" + k: scene[k] for k in SCENE_KEYS}
    d.update({"in_" + k: cam[k] for k in CAM_KEYS})
    if frames is not None:
        d["in_frames_pos_scale"] = np.stack([f[0] for f in frames])
        d["in_frames_rot"] = np.stack([f[1] for f in frames])
        d["in_frames_dirty"] = np.stack([f[2] for f in frames])
    d.update({"ref_" + k: v for k, v in ref.items()})
    save(name, **d)


def attach_fixture(name):
    """Joint attachments (model.c:1626-1641) + the camera bounding-volume pick (model.c:1703-1713)."""
    rng = np.random.Generator(np.random.PCG64(3))
    scene = synth.pad_levels(synth.entities_forest(900, seed=13, n_models=3))
    scene["model_skip"][:] = 0
    kids = np.flatnonzero((scene["parent"] >= 0) & (scene["orig_of"] >= 0))
    ent = np.sort(rng.choice(kids, 30, replace=False)).astype(np.uint32)
    jt, bind = synth._rigid_mat4(rng, 30, 1.0), synth._rigid_mat4(rng, 30, 1.0)
    joint = (np.arange(30) + 7).astype(np.int32)           # never JOINT_TYPE_MAX (6): the reference's "no joint"
    cam = synth.camera(pos=(0, 5, 60))
    plain = refrun.entities(scene, cam)
    real = np.flatnonzero((scene["orig_of"] >= 0) & ((scene["flags"] & synth.E_ALIVE) != 0))
    cam_pos = plain["center"][0][real[50]]                 # inside entity real[50]'s box
    inside_of = real[120]                                  # the control entity (its own box is excluded)
    frames = multi_frame(scene, seed=5, n_frames=3)
    ref = refrun.entities(scene, cam, frames, attach=dict(entity=ent, joint=joint, jt=jt, bind=bind),
                          bv=dict(cam_pos=cam_pos, ctl=int(inside_of)))
    d = {"in_" + k: scene[k] for k in SCENE_KEYS}
    d.update({"in_" + k: cam[k] for k in CAM_KEYS})
    d.update(in_frames_pos_scale=np.stack([f[0] for f in frames]), in_frames_rot=np.stack([f[1] for f in frames]),
             in_frames_dirty=np.stack([f[2] for f in frames]), in_attach_entity=ent, in_attach_jt=jt, in_attach_bind=bind,
             in_bv_cam_pos=np.asarray(cam_pos, np.float32), in_bv_ctl=np.asarray([inside_of], np.int32))
    d.update({"ref_" + k: v for k, v in ref.items()})
    save(name, **d)


def multi_frame(scene, seed, n_frames=3, dirty_frac=0.2):
    """Frames after the first move a random subset (exercises seq/parent_seq skipping)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(scene["n"])
    frames = [(scene["pos_scale"].copy(), scene["rot"].copy(), np.ones(n, np.uint8))]
    for _ in range(n_frames - 1):
        ps, rot, _d = (a.copy() for a in frames[-1])
        dirty = (rng.uniform(0, 1, n) < dirty_frac).astype(np.uint8)
        idx = np.flatnonzero(dirty)
        ps[idx, :3] += rng.uniform(-1, 1, (len(idx), 3)).astype(np.float32)
        ang = rng.uniform(-3, 3, (len(idx), 3))
        rot[idx] = synth.quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2])
        frames.append((ps, rot, dirty))
    return frames


def particle_fixture(name, ps, frames, view_mx, state=synth.DRAND48_DEFAULT_STATE):
    ref = refrun.particles(ps, view_mx, state, frames)
    save(name, in_sys=ps["sys"], in_row_sys=ps["row_sys"], in_n=np.asarray([ps["n"], ps["n_real"]], np.uint32),
         in_view_mx=np.asarray(view_mx, np.float32), in_rng_state=np.asarray([state], np.uint64),
         **{"ref_" + k: v for k, v in ref.items()})


SK_KEYS = ("parent", "invmx", "root_pose", "order")
AN_KEYS = ("ch_target", "ch_path", "ch_nr", "ch_time_off", "ch_data_off", "times", "data")


def pose_fixture(name, sk, an, chars, char_times):
    ref = refrun.pose(sk, an, chars, char_times)
    d = {"sk_" + k: sk[k] for k in SK_KEYS}
    d.update({"an_" + k: an[k] for k in AN_KEYS})
    d.update(in_char_mx=chars["char_mx"], in_trs0=chars["trs0"], in_char_times=np.asarray(char_times, np.float32))
    d.update({"ref_" + k: v for k, v in ref.items() if k != "time_end"})
    d["ref_time_end"] = np.asarray([ref["time_end"]], np.float32)
    save(name, **d)


def light_fixture(name, lights, cam, width, height, cell):
    tiles, radius, view_mx, proj_mx = refrun.lightgrid(lights, cam, width, height, cell)
    d = {"in_" + k: np.asarray(lights[k]) for k in ("pos", "color", "attenuation", "is_dir", "active")}
    d.update({"cam_" + k: cam[k] for k in CAM_KEYS})
    d.update(in_grid=np.asarray([width, height, cell], np.uint32), ref_tiles=tiles, ref_radius=radius,
             ref_view_mx=view_mx, ref_proj_mx=proj_mx)
    save(name, **d)


def character_fixture(name, n=600, seed=13, frames=4):
    feed = synth.character_feed(n, seed=seed, with_bodies=False)
    rng = np.random.Generator(np.random.PCG64(seed))
    rot = synth.quat_from_euler_xyz(*rng.uniform(-3, 3, (3, n))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, n).astype(np.float32)
    pos_frames = np.stack([feed["pos"] + np.float32(f) * np.asarray([0, -30, 0], np.float32) for f in range(frames)])
    ref = refrun.characters(pos_frames, rot, scale, feed["hist_pos"], feed["hist_head"], feed["hist_wrapped"],
                            feed["limbo_height"])
    save(name, in_pos_frames=pos_frames.astype(np.float32), in_rot=rot, in_scale=scale, in_hist_pos=feed["hist_pos"],
         in_hist_head=feed["hist_head"], in_hist_wrapped=feed["hist_wrapped"],
         in_limbo_height=np.asarray([feed["limbo_height"]], np.float32), **{"ref_" + k: v for k, v in ref.items()})


def transform_fixture(name, n=400, seed=21):
    rng = np.random.Generator(np.random.PCG64(seed))
    degrees = (rng.uniform(0, 1, n) < 0.5).astype(np.uint8)
    angles = np.where(degrees[:, None] != 0, rng.uniform(-400, 400, (n, 3)), rng.uniform(-7, 7, (n, 3))).astype(np.float32)
    angles[:6] = [[0, 0, 0], [180, -180, 180.00002], [np.pi, -np.pi, 3.1415927], [360, 0, -360], [0, 90, 0], [1e-30, 0, 0]]
    degrees[:6] = [0, 1, 0, 1, 1, 0]
    pos = rng.uniform(-500, 500, (n, 3)).astype(np.float32)
    off = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    quat, moved = refrun.transform(angles, degrees, pos, off)
    save(name, in_angles=angles, in_degrees=degrees, in_pos=pos, in_off=off, ref_quat=quat, ref_pos=moved)


def clock_fixture(name, n=12, J=24, frames=8, seed=8):
    sk = synth.skeleton(J, 6, seed=seed)
    an = synth.animation(J, 9, 2.0, seed=seed)
    ch = synth.characters(n, J, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    clock = dict(start=rng.uniform(100.0, 101.0, n), speed=rng.choice([0.5, 1.0, 1.7, 3.0], n).astype(np.float32),
                 repeat=np.ones(n, np.uint8), now=101.0 + np.cumsum(rng.uniform(0.05, 0.5, frames)))
    ref = refrun.pose(sk, an, ch, clock=clock)
    d = {"sk_" + k: sk[k] for k in SK_KEYS}
    d.update({"an_" + k: an[k] for k in AN_KEYS})
    d.update(in_char_mx=ch["char_mx"], in_trs0=ch["trs0"])
    d.update({"clock_" + k: v for k, v in clock.items()})
    d.update(ref_ani_time=ref["ani_time"], ref_joint_transforms=ref["joint_transforms"],
             ref_time_end=np.asarray([ref["time_end"]], np.float32))
    save(name, **d)


def main():
    if not refrun.available():
        refrun.build()
    os.makedirs(OUT, exist_ok=True)
    cam = synth.camera()
    from oracle import binding as ob
    _fr, view, _proj = ob.frustum_from_camera(synth.camera(pos=(1, 2, 3), quat=synth.quat_from_euler_xyz(0.1, 0.7, -0.2)))
    # view_mx comes from the oracle here only as an INPUT matrix; outputs are the reference's
    particle_fixture("particles_uniform", synth.particle_systems(n_sys=6, count=200, radius=3.0, velocity=0.5), 5, view)
    particle_fixture("particles_ragged", synth.particle_systems(n_sys=9, count=150, radius=4.0, velocity=0.4,
                                                                ragged=True, seed=5), 6, view, state=0x0BADC0FFEE42)
    for dist, nm in ((synth.PART_DIST_CBRT, "cbrt"), (synth.PART_DIST_POW075, "pow075")):
        particle_fixture("particles_" + nm, synth.particle_systems(n_sys=3, count=100, radius=2.0, velocity=0.6,
                                                                    dist=dist, seed=6), 4, view)
    for nm, skw, akw in (("pose_full", dict(), dict()),
                         ("pose_ragged", dict(unreachable=4), dict(missing_frac=0.3, ragged=True))):
        sk = synth.skeleton(24, 6, seed=8, **skw)
        an = synth.animation(24, 9, 2.0, seed=8, **akw)
        ch = synth.characters(12, 24, seed=8)
        t = np.stack([ch["phase"], (ch["phase"] * 1.7) % 2.3, np.full(12, 2.0, np.float32),
                      np.full(12, -0.25, np.float32), np.zeros(12, np.float32)]).astype(np.float32)
        pose_fixture(nm, sk, an, ch, t)
    attach_fixture("attach_bv_frames")
    character_fixture("characters_limbo")
    clock_fixture("animclock_frames")
    transform_fixture("transform_verbs")
    tilted = synth.camera(pos=(1.0, 2.0, 3.0), quat=synth.quat_from_euler_xyz(0.1, 0.2, -0.05))
    light_fixture("lightgrid_1080p", synth.lights(seed=7), tilted, 1920, 1080, synth.LIGHT_TILE)
    light_fixture("lightgrid_odd_z01", synth.lights(97, seed=8, n_dir=1, inactive_frac=0.3),
                  synth.camera(pos=(-4, 1, 9), quat=synth.quat_from_euler_xyz(-0.2, 2.8, 0.1), ndc_z_zero_one=1),
                  333, 222, 16)
    entity_fixture("entities_flat_c1", synth.entities_flat(512, seed=1234), cam)
    entity_fixture("entities_flat_euler", synth.entities_flat(512, seed=99, full_euler=True),
                   synth.camera(pos=(10, 5, -20), quat=synth.quat_from_euler_xyz(0.2, 2.5, -0.1),
                                ndc_z_zero_one=1))
    entity_fixture("entities_chains", synth.entities_chains(96, 8, seed=2), cam)
    forest = synth.pad_levels(synth.entities_forest(700, seed=7))
    forest_unpadded = {k: v for k, v in forest.items()}
    ref_frames = multi_frame(forest, seed=11)
    # forest is already level-padded: pad_levels() inside entity_fixture is then the identity
    entity_fixture("entities_forest_frames", forest_unpadded, synth.camera(pos=(0, 10, 50)), ref_frames)


if __name__ == "__main__":
    main()
