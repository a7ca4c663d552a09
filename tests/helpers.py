"""Shared helpers of the parity tests."""
import numpy as np

from clap_amd import synth

SCENE_KEYS = ("pos_scale", "rot", "parent", "model", "model_aabb", "model_skip", "flags", "level_start")
CAM_KEYS = ("cam_pos", "cam_quat", "persp", "ndc_z_zero_one")


def bits_equal(a, b):
    """Bit-exact comparison of float arrays (distinguishes -0.0 from 0.0, compares NaN payloads)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def assert_bits_equal(a, b, what):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    bad = np.flatnonzero(a.view(np.uint32).ravel() != b.view(np.uint32).ravel())
    assert bad.size == 0, (f"{what}: {bad.size} of {a.size} floats differ bitwise; first at flat index "
                           f"{bad[0]}: {a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}")


def load_golden(path):
    z = np.load(path)
    scene = {k: z["in_" + k] for k in SCENE_KEYS}
    scene["n"] = scene["parent"].shape[0]
    scene["seqs"] = np.zeros(scene["n"], np.uint32)
    cam = {k: z["in_" + k] for k in CAM_KEYS}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    frames = None
    if "in_frames_pos_scale" in z.files:
        frames = list(zip(z["in_frames_pos_scale"], z["in_frames_rot"], z["in_frames_dirty"]))
    return scene, cam, ref, frames


def apply_frame(scene, st, frame):
    """Host-side mutation a game would do between frames: write TRS of dirty entities, set dirty."""
    ps, rot, dirty = frame
    idx = np.flatnonzero(dirty)
    scene["pos_scale"][idx] = ps[idx]
    scene["rot"][idx] = rot[idx]
    st["flags"][idx] |= synth.E_DIRTY
    return idx
