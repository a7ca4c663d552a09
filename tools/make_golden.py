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
