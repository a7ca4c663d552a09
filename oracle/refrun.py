"""Runner for oracle/_ref/clap_ref (the real reference code) -- TEST INFRASTRUCTURE ONLY.

The binary is built from /root/reference in the build container
(``make -C oracle/ref``) and travels to the GPU box prebuilt; it is absent
wherever it was never built, and callers must check ``available()``.
"""
import os
import subprocess
import tempfile

import numpy as np

from . import clpio

_HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(_HERE, "_ref", "clap_ref")
REF_ROOT = "/root/reference"


def available():
    return os.path.exists(BIN) and os.access(BIN, os.X_OK)


def build():
    """(Re)build from /root/reference when it is present; no-op otherwise."""
    if os.path.isdir(os.path.join(REF_ROOT, "core")):
        subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "ref")], check=True)


def run(command, arrays, timeout=600):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.clpio"), os.path.join(td, "out.clpio")
        clpio.write(fin, arrays)
        subprocess.run([BIN, command, fin, fout], check=True, timeout=timeout)
        return clpio.read(fout)


def _camera_arrays(cam):
    return {k: cam[k] for k in ("cam_pos", "cam_quat", "persp", "ndc_z_zero_one")}


def entities(scene, cam, frames=None, attach=None, bv=None):
    """Run default_update x frames + view_entity_in_frustum on the reference.

    frames: list of (pos_scale, rot, dirty) per frame; default = one frame, the
    scene's own arrays with every entity dirty.
    attach: dict(entity u32[k], joint i32[k], jt f32[k,16], bind f32[k,16]) joint attachments.
    bv: dict(cam_pos f32[3], ctl int) -> also returns the camera bounding-volume pick per frame.
    """
    n = int(scene["n"])
    if frames is None:
        frames = [(scene["pos_scale"], scene["rot"], np.ones(n, np.uint8))]
    f = len(frames)
    arrays = dict(n=np.asarray([n], np.uint32), frames=np.asarray([f], np.uint32),
                  pos_scale=np.stack([fr[0] for fr in frames]).astype(np.float32),
                  rot=np.stack([fr[1] for fr in frames]).astype(np.float32),
                  dirty=np.stack([fr[2] for fr in frames]).astype(np.uint8),
                  parent=scene["parent"], model=scene["model"], model_aabb=scene["model_aabb"],
                  model_skip=scene["model_skip"], flags=scene["flags"])
    arrays.update(_camera_arrays(cam))
    if attach is not None:
        arrays.update(attach_entity=np.asarray(attach["entity"], np.uint32), attach_joint=np.asarray(attach["joint"], np.int32),
                      attach_jt=np.asarray(attach["jt"], np.float32), attach_bind=np.asarray(attach["bind"], np.float32))
    if bv is not None:
        arrays.update(bv_cam_pos=np.asarray(bv["cam_pos"], np.float32), bv_ctl=np.asarray([bv["ctl"]], np.int32))
    out = run("entities", arrays)
    A = clpio.as_array
    extra = {}
    if bv is not None:
        extra = dict(bv=A(out["bv"], np.int32), bv_volume=A(out["bv_volume"], np.float32))
    return dict(**extra, mx=A(out["mx"], np.float32, (f, n, 16)), inv_mx=A(out["inv_mx"], np.float32, (f, n, 16)),
                aabb=A(out["aabb"], np.float32, (f, n, 6)), center=A(out["center"], np.float32, (f, n, 3)),
                seqs=A(out["seqs"], np.uint32, (f, n)), visible=A(out["visible"], np.uint8, (f, n)),
                view_mx=A(out["view_mx"], np.float32), proj_mx=A(out["proj_mx"], np.float32),
                planes=A(out["planes"], np.float32, (6, 4)), corners=A(out["corners"], np.float32, (8, 4)))


def bench_entities(scene, cam, reps=5):
    n = int(scene["n"])
    arrays = dict(n=np.asarray([n], np.uint32), reps=np.asarray([reps], np.uint32),
                  pos_scale=scene["pos_scale"], rot=scene["rot"], parent=scene["parent"],
                  model=scene["model"], model_aabb=scene["model_aabb"], model_skip=scene["model_skip"],
                  flags=scene["flags"])
    arrays.update(_camera_arrays(cam))
    out = run("bench_entities", arrays)
    best, mean, visible = clpio.as_array(out["seconds"], np.float64)
    return dict(best_s=float(best), mean_s=float(mean), visible=int(visible))


def particles(ps, view_mx, rng_state, frames):
    """particle_spawn x count then particles_update x frames on the reference, systems in
    order, one libc drand48 stream seeded to `rng_state`.  Arrays come back in the padded
    particle layout of synth.particle_systems (padding rows zero)."""
    sys = ps["sys"]
    ns = sys.shape[0]
    arrays = dict(n_sys=np.asarray([ns], np.uint32), frames=np.asarray([frames], np.uint32),
                  center=sys["center"].astype(np.float32), dist=sys["dist"].astype(np.uint32),
                  radius=sys["radius"].astype(np.float64), min_radius=sys["min_radius"].astype(np.float64),
                  velocity=sys["velocity"].astype(np.float64), count=sys["count"].astype(np.uint32),
                  view_mx=np.asarray(view_mx, np.float32), rng_state=np.asarray([rng_state], np.uint64))
    out = run("particles", arrays)
    total = int(sys["count"].sum())
    A = clpio.as_array
    n = int(ps["n"])
    idx = np.concatenate([np.arange(int(f), int(f) + int(c)) for f, c in zip(sys["first"], sys["count"])])

    def pad(a):          # packed [.., total, 3] -> padded [.., n, 3]
        o = np.zeros(a.shape[:-2] + (n, 3), np.float32)
        o[..., idx, :] = a
        return o

    return dict(pos0=pad(A(out["pos0"], np.float32, (total, 3))), vel0=pad(A(out["vel0"], np.float32, (total, 3))),
                pos=pad(A(out["pos"], np.float32, (frames, total, 3))),
                vel=pad(A(out["vel"], np.float32, (frames, total, 3))),
                mx=A(out["mx"], np.float32, (frames, ns, 16)),
                rng_state=A(out["rng_state"], np.uint64, (frames + 1,)))


def pose(sk, an, chars, char_times=None, clock=None):
    """channels_transform + one_joint_transform on the reference for every character;
    char_times [frames, n_chars] (float32 frame times).  With clock = dict(now f64[frames],
    start f64[n], speed f32[n], repeat u8[n]) the characters are driven through animated_update
    (queue entry pushed at `start`, then one call per `now`) and ani_time per frame is returned too."""
    if clock is not None:
        char_times = np.zeros((len(clock["now"]), len(clock["start"])), np.float32)
    char_times = np.atleast_2d(np.asarray(char_times, np.float32))
    frames, n = char_times.shape
    J = int(sk["nr_joints"])
    arrays = dict(nr_joints=np.asarray([J], np.uint32), n_chars=np.asarray([n], np.uint32),
                  frames=np.asarray([frames], np.uint32), n_channels=np.asarray([an["n_channels"]], np.uint32),
                  parent=sk["parent"], invmx=sk["invmx"], root_pose=sk["root_pose"],
                  ch_target=an["ch_target"], ch_path=an["ch_path"], ch_nr=an["ch_nr"],
                  ch_time_off=an["ch_time_off"], ch_data_off=an["ch_data_off"], times=an["times"], data=an["data"],
                  char_time=char_times, char_mx=chars["char_mx"][:n], trs0=chars["trs0"])
    if clock is not None:
        arrays.update(now=np.asarray(clock["now"], np.float64), start=np.asarray(clock["start"], np.float64),
                      speed=np.asarray(clock["speed"], np.float32), repeat=np.asarray(clock["repeat"], np.uint8))
    out = run("pose", arrays)
    A = clpio.as_array
    extra = dict(ani_time=A(out["ani_time"], np.float64, (frames, n))) if clock is not None else {}
    return dict(**extra, trs=A(out["trs"], np.float32, (frames, n, J, 10)),
                joint_transforms=A(out["joint_transforms"], np.float32, (frames, n, J, 16)),
                globalmx=A(out["global"], np.float32, (frames, n, J, 16)),
                joint_pos=A(out["joint_pos"], np.float32, (frames, n, J, 4)),
                bind=A(out["bind"], np.float32, (J, 16)), time_end=float(A(out["time_end"], np.float32)[0]))


def lod_blocks(model_aabb, scale, lod_min, lod_max, request):
    """entity3d_aabb_avg_edge and entity3d_set_lod(e, request, false) of the reference per probe."""
    n = len(scale)
    out = run("lod", dict(n=np.asarray([n], np.uint32), model_aabb=np.asarray(model_aabb, np.float32),
                          scale=np.asarray(scale, np.float32), lod_min=np.asarray(lod_min, np.int32),
                          lod_max=np.asarray(lod_max, np.int32), request=np.asarray(request, np.int32)))
    return clpio.as_array(out["avg_edge"], np.float32), clpio.as_array(out["cur_lod"], np.int32)


def lightgrid(lights, cam, width, height, cell):
    """The reference's light_grid_compute: returns (tiles u32[th][tw][4], radius f32[nr], view_mx, proj_mx)."""
    nr = int(lights["nr_lights"])
    arrays = dict(nr_lights=np.asarray([nr], np.uint32), grid=np.asarray([width, height, cell], np.uint32),
                  active=np.asarray(lights["active"], np.uint32), is_dir=np.asarray(lights["is_dir"], np.int32),
                  pos=np.asarray(lights["pos"], np.float32), color=np.asarray(lights["color"], np.float32),
                  attenuation=np.asarray(lights["attenuation"], np.float32))
    arrays.update(_camera_arrays(cam))
    out = run("lightgrid", arrays)
    tw, th = (int(v) for v in clpio.as_array(out["tile_dims"], np.uint32))
    tiles = clpio.as_array(out["tiles"], np.uint32).reshape(th, tw, 4)
    return (tiles, clpio.as_array(out["radius"], np.float32), clpio.as_array(out["view_mx"], np.float32),
            clpio.as_array(out["proj_mx"], np.float32))


def characters(pos_frames, rot, scale, hist_pos, hist_head, hist_wrapped, limbo_height):
    """The reference's character_update (+ chained default_update) for body-less characters, one call per
    frame after entity3d_position(pos_frames[f]).  Returns dict(pos, mx, hist_head, hist_wrapped) per frame."""
    pos_frames = np.ascontiguousarray(pos_frames, np.float32)
    f, n = pos_frames.shape[:2]
    out = run("characters", dict(n=np.asarray([n], np.uint32), frames=np.asarray([f], np.uint32),
                                 limbo_height=np.asarray([limbo_height], np.float32), pos=pos_frames,
                                 rot=np.asarray(rot, np.float32), scale=np.asarray(scale, np.float32),
                                 hist_pos=np.asarray(hist_pos, np.float32), hist_head=np.asarray(hist_head, np.uint32),
                                 hist_wrapped=np.asarray(hist_wrapped, np.uint8)))
    return dict(pos=clpio.as_array(out["pos"], np.float32).reshape(f, n, 3),
                mx=clpio.as_array(out["mx"], np.float32).reshape(f, n, 16),
                hist_head=clpio.as_array(out["hist_head"], np.uint32).reshape(f, n),
                hist_wrapped=clpio.as_array(out["hist_wrapped"], np.uint8).reshape(f, n))


def transform(angles, degrees, pos, off):
    """transform_set_pos + transform_move + transform_set_angles of the reference -> (quat xyzw, pos)."""
    n = len(angles)
    out = run("transform", dict(n=np.asarray([n], np.uint32), angles=np.asarray(angles, np.float32),
                                degrees=np.asarray(degrees, np.uint8), pos=np.asarray(pos, np.float32),
                                off=np.asarray(off, np.float32)))
    return clpio.as_array(out["quat"], np.float32).reshape(n, 4), clpio.as_array(out["pos"], np.float32).reshape(n, 3)
