"""GPU: pose blend + joint palette + vertex skinning (through the C ABI) against the oracle and
the reference's golden vectors.

Tolerance (BASELINE.json north_star: "within 1e-5 relative for float transforms/positions"):
the reference evaluates lerp in double and slerp through double acos/sin/cos; device libm is not
glibc, so these outputs are compared with rtol 1e-5 on a scale of the array's own magnitude.
"""
import glob
import os

import numpy as np
import torch
import pytest

from clap_amd import synth
from oracle import binding as ob
from test_oracle_pose import load_pose

pytestmark = pytest.mark.gpu
RTOL = 1e-5

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pose_*.npz")))


def assert_close(got, exp, what, scale=None):
    got = np.asarray(got, np.float64)
    exp = np.asarray(exp, np.float64)
    assert got.shape == exp.shape, what
    s = max(float(np.abs(exp).max()), 1e-30) if scale is None else scale
    err = float(np.abs(got - exp).max()) / s
    assert err <= RTOL, f"{what}: max |diff| / max |ref| = {err:.3e} > {RTOL}"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_pose_matches_reference_golden(path, cuda_device):
    from clap_amd import animation
    sk, an, ch, times, ref = load_pose(path)
    n, J = ch["char_mx"].shape[0], sk["nr_joints"]
    model = animation.SkinnedModel(sk, [an], bind=ref["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    reach = sk["order"]
    unreach = np.setdiff1d(np.arange(J), reach)
    for f in range(times.shape[0]):
        batch.set_frame_times(times[f])
        batch.pose_update()
        out = batch.download()
        assert_close(out["trs"], ref["trs"][f], f"frame {f} T/R/S")
        assert_close(out["joint_transforms"][:, reach], ref["joint_transforms"][f][:, reach], f"frame {f} joint_transforms")
        assert_close(out["joint_pos"][:, reach], ref["joint_pos"][f][:, reach], f"frame {f} joint pos")
        assert not out["joint_transforms"][:, unreach].any(), "joints outside joint 0's tree must stay untouched"


@pytest.mark.parametrize("J,depth,skw,akw", [
    (64, 8, {}, {}),                                                   # BASELINE config 3 skeleton
    (64, 8, dict(unreachable=6), dict(missing_frac=0.25, ragged=True)),
    (5, 3, {}, {}),
    (100, 12, {}, dict(ragged=True)),                                  # two wavefronts per character
    (200, 20, dict(unreachable=3), dict(missing_frac=0.1)),            # JOINTS_MAX
], ids=["c3_64j", "ragged_64j", "tiny_5j", "two_wave_100j", "joints_max_200j"])
def test_pose_matches_oracle(J, depth, skw, akw, cuda_device):
    from clap_amd import animation
    sk = synth.skeleton(J, depth, seed=11, **skw)
    an0 = synth.animation(J, 30, 2.0, seed=11, **akw)
    an1 = synth.animation(J, 7, 1.0, seed=12)
    n = 37
    ch = synth.characters(n, J, seed=11)
    sk["bind"] = ob.skeleton_bind(sk)
    model = animation.SkinnedModel(sk, [an0, an1], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    which = (np.arange(n) % 3 == 0).astype(np.int32)                   # a third of the characters play animation 1
    trs = np.tile(ch["trs0"], (n, 1, 1))
    reach = sk["order"]
    for f, t in enumerate([ch["phase"], (ch["phase"] + 0.61) % 2.4, np.full(n, 2.0, np.float32), np.full(n, -1.0, np.float32)]):
        t = t.astype(np.float32)
        jt = np.zeros((n, J, 16), np.float32)
        jp = np.zeros((n, J, 4), np.float32)
        for a_id, an in ((0, an0), (1, an1)):
            sel = np.flatnonzero(which == a_id)
            sub = trs[sel].copy()
            j1, _g, p1 = ob.pose(sk, an, t[sel], ch["char_mx"][sel], sub)
            trs[sel], jt[sel], jp[sel] = sub, j1, p1
        batch.set_frame_times(t, which)
        batch.pose_update()
        out = batch.download()
        assert_close(out["trs"], trs, f"frame {f} T/R/S")
        assert_close(out["joint_transforms"][:, reach], jt[:, reach], f"frame {f} joint_transforms")
        assert_close(out["joint_pos"][:, reach], jp[:, reach], f"frame {f} joint pos")


@pytest.mark.parametrize("J", [64, 40, 1], ids=["64j", "40j_short_rows", "one_joint"])
def test_pose_streaming_loop_clips_rows_masks_and_tail(J, cuda_device):
    """The one-wavefront-per-character loop (every joint animated on all paths and reachable) stores through buffer
    descriptors that clip rows shorter than 64 joints, the characters past the end of the last group and masked
    outputs: the arrays carry guard values around and inside them that must survive, characters map to entity
    matrices through an index list, key counts are ragged (incl. channels of two keys) and the times reach
    before the first / past the last key."""
    from clap_amd import animation
    n = 1003                                                        # not a multiple of the 4 characters of a block
    sk = synth.skeleton(J, min(8, J), seed=31)
    an0 = synth.animation(J, 30, 2.0, seed=31, ragged=True)
    an1 = synth.animation(J, 2, 1.0, seed=32)
    ch = synth.characters(n, J, seed=31)
    sk["bind"] = ob.skeleton_bind(sk)
    rng = np.random.default_rng(5)
    perm = rng.permutation(n + 7).astype(np.uint32)[:n]             # character -> entity (matrices of n + 7 entities)
    ent_mx = rng.standard_normal((n + 7, 16)).astype(np.float32)
    model = animation.SkinnedModel(sk, [an0, an1], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ent_mx, entity_index=perm)
    which = (np.arange(n) % 4 == 1).astype(np.int32)
    t = ((ch["phase"] * 1.3) - 0.2).astype(np.float32)              # some below 0, some past an1's end
    trs = np.tile(ch["trs0"], (n, 1, 1))
    jt = np.zeros((n, J, 16), np.float32)
    jp = np.zeros((n, J, 4), np.float32)
    for a_id, an in ((0, an0), (1, an1)):
        sel = np.flatnonzero(which == a_id)
        sub = trs[sel].copy()
        j1, _g, p1 = ob.pose(sk, an, t[sel], ent_mx[perm[sel]], sub)
        trs[sel], jt[sel], jp[sel] = sub, j1, p1
    batch.set_frame_times(t, which)
    reach = sk["order"]
    for trs_on, pos_on in ((True, True), (False, True), (True, False), (False, False)):
        batch.trs.fill_(7.5); batch.joint_transforms.fill_(-3.25); batch.joint_pos.fill_(9.0)
        batch.trs.copy_(torch.from_numpy(np.tile(ch["trs0"], (n, 1, 1))))
        if not trs_on:
            batch.trs.fill_(7.5)
        batch.set_outputs(trs=trs_on, joint_pos=pos_on)
        batch.pose_update()
        out = batch.download()
        what = f"J={J} trs={trs_on} pos={pos_on}"
        if trs_on:
            assert_close(out["trs"], trs, what + " T/R/S")
        else:
            assert (out["trs"] == 7.5).all(), what + ": masked T/R/S was written"
        assert_close(out["joint_transforms"][:, reach], jt[:, reach], what + " joint_transforms")
        if pos_on:
            assert_close(out["joint_pos"][:, reach], jp[:, reach], what + " joint pos")
        else:
            assert (out["joint_pos"] == 9.0).all(), what + ": masked joint positions were written"


@pytest.mark.parametrize("akw,two", [({}, False), (dict(ragged=True, missing_frac=0.1), True)], ids=["c3_pool", "two_ragged_anims"])
def test_pose_many_characters_matches_oracle(akw, two, cuda_device):
    """9000 characters = 2250 character groups over at most 1024 resident blocks: the persistent loop,
    its one-character-ahead input requests and the staging tile reuse are all exercised against the oracle."""
    from clap_amd import animation
    J, n = 64, 9000
    sk = synth.skeleton(J, 8, seed=21, unreachable=2)
    anims = [synth.animation(J, 30, 2.0, seed=21, **akw)] + ([synth.animation(J, 7, 1.0, seed=22)] if two else [])
    ch = synth.characters(n, J, seed=21)
    sk["bind"] = ob.skeleton_bind(sk)
    model = animation.SkinnedModel(sk, anims, bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    which = (np.arange(n) % 3 == 0).astype(np.int32) if two else np.zeros(n, np.int32)
    trs = np.tile(ch["trs0"], (n, 1, 1))
    reach = sk["order"]
    for f, t in enumerate([ch["phase"], (ch["phase"] * 1.37 + 0.2) % 2.2]):
        t = t.astype(np.float32)
        jt = np.zeros((n, J, 16), np.float32)
        jp = np.zeros((n, J, 4), np.float32)
        for a_id, an in enumerate(anims):
            sel = np.flatnonzero(which == a_id)
            sub = trs[sel].copy()
            j1, _g, p1 = ob.pose(sk, an, t[sel], ch["char_mx"][sel], sub)
            trs[sel], jt[sel], jp[sel] = sub, j1, p1
        batch.set_frame_times(t, which)
        batch.pose_update()
        out = batch.download()
        assert_close(out["trs"], trs, f"frame {f} T/R/S")
        assert_close(out["joint_transforms"][:, reach], jt[:, reach], f"frame {f} joint_transforms")
        assert_close(out["joint_pos"][:, reach], jp[:, reach], f"frame {f} joint pos")


def test_animated_update_clock_on_device(cuda_device, golden_dir):
    """animated_update's clock (model.c:1563-1592) as a kernel: frame times, `ended`, restart of
    repeating entries -- against the reference's own animated_update (golden) and the oracle."""
    import os
    from clap_amd import animation
    z = np.load(os.path.join(golden_dir, "animclock_frames.npz"))
    sk = {k[3:]: z[k] for k in z.files if k.startswith("sk_")}
    sk["nr_joints"] = sk["parent"].shape[0]
    an = {k[3:]: z[k] for k in z.files if k.startswith("an_")}
    an["n_channels"] = an["ch_target"].shape[0]
    an["time_end"] = float(z["ref_time_end"][0])
    sk["bind"] = ob.skeleton_bind(sk)
    n = len(z["clock_start"])
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, z["in_trs0"], z["in_char_mx"])
    batch.start_clock(ani_time=z["clock_start"], speed=z["clock_speed"], repeat=z["clock_repeat"])
    ani = z["clock_start"].astype(np.float64).copy()
    te = np.asarray([an["time_end"]], np.float32)
    reach = sk["order"]
    restarts = 0
    for f, now in enumerate(z["clock_now"]):
        ft, ended = ob.animation_time(np.zeros(n, np.uint32), te, ani, z["clock_speed"], z["clock_repeat"], now)
        batch.animated_update(now)
        clk = batch.download_clock()
        assert np.array_equal(clk["ani_time"].view(np.uint64), z["ref_ani_time"][f].view(np.uint64)), f"frame {f} ani_time"
        assert np.array_equal(clk["ani_time"], ani) and np.array_equal(clk["ended"], ended)
        assert np.array_equal(clk["frame_time"].view(np.uint32), ft.view(np.uint32)), f"frame {f} frame_time"
        assert_close(batch.download()["joint_transforms"][:, reach], z["ref_joint_transforms"][f][:, reach], f"frame {f}")
        restarts += int(ended.sum())
    assert restarts > n // 2, "the fixture runs past the end of the animation"


def test_animation_clock_non_repeating_entries_are_left_to_the_host(cuda_device):
    from clap_amd import animation
    sk = synth.skeleton(16, 4, seed=5)
    an = synth.animation(16, 6, 1.5, seed=5)
    ch = synth.characters(64, 16, seed=5)
    sk["bind"] = ob.skeleton_bind(sk)
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, 64, ch["trs0"], ch["char_mx"])
    repeat = (np.arange(64) % 2).astype(np.uint8)
    start = np.linspace(0, 1, 64)
    batch.start_clock(ani_time=start, speed=np.full(64, 1.25, np.float32), repeat=repeat)
    ani = start.copy()
    te = np.asarray(model.time_end, np.float32)
    for now in (1.0, 1.7, 2.9, 3.0):
        ft, ended = ob.animation_time(np.zeros(64, np.uint32), te, ani, np.full(64, 1.25, np.float32), repeat, now)
        batch.animated_update(now)
        clk = batch.download_clock()
        assert np.array_equal(clk["ani_time"], ani) and np.array_equal(clk["ended"], ended)
    assert np.array_equal(ani[repeat == 0], start[repeat == 0]), "no restart without `repeat`"
    assert (ani[repeat == 1] != start[repeat == 1]).any()


@pytest.mark.parametrize("shared_mesh", [True, False], ids=["instanced_mesh", "mesh_per_character"])
def test_skin_matches_oracle(shared_mesh, cuda_device):
    from clap_amd import animation
    J, n = 64, 23
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    sk["bind"] = ob.skeleton_bind(sk)
    if shared_mesh:
        mesh = synth.skinned_mesh(200, J, seed=3)
        vf, vc = np.zeros(n, np.uint32), np.full(n, 200, np.uint32)
    else:                                                             # ragged: 1 .. 700 vertices per character
        vc = np.random.Generator(np.random.PCG64(4)).integers(1, 700, n).astype(np.uint32)
        mesh = synth.skinned_mesh(int(vc.sum()), J, seed=3)
        vf = np.concatenate([[0], np.cumsum(vc[:-1])]).astype(np.uint32)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
    batch.set_frame_times(ch["phase"])
    batch.pose_update()
    batch.skin()
    out = batch.download()
    exp_p, exp_n = ob.skin(mesh, vf, vc, out["joint_transforms"])     # same palette in: isolates the skinning kernel
    assert_close(out["out_position"], exp_p, "skinned positions")
    assert_close(out["out_normal"], exp_n, "skinned normals")
    # and end to end against the oracle's own palette
    trs = np.tile(ch["trs0"], (n, 1, 1))
    jt, _g, _p = ob.pose(sk, an, ch["phase"], ch["char_mx"], trs)
    exp_p2, _ = ob.skin(mesh, vf, vc, jt)
    assert_close(out["out_position"], exp_p2, "pose -> skin end to end")


def test_c3_shape_properties(cuda_device):
    """BASELINE config 3 at reduced count (2k characters here; bench.py runs the 50k): palette of a
    rest pose equals I (global * invmx with global == bind), skinning by it is the identity."""
    from clap_amd import animation
    J, n = 64, 2000
    sk = synth.skeleton(J, 8, seed=3)
    bind = ob.skeleton_bind(sk)
    sk["bind"] = bind
    # an "animation" whose keys hold each joint at its bind pose is not expressible without the
    # local decomposition; use linearity instead: two palettes P1, P2 -> skin is linear in the palette
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    mesh = synth.skinned_mesh(200, J, seed=3)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, bind=bind, device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    batch.set_frame_times(ch["phase"])
    batch.pose_update()
    batch.skin()
    o1 = batch.download()
    batch.joint_transforms.mul_(2.0)                                  # linearity: 2 * palette -> 2 * output
    batch.skin()
    o2 = batch.download()
    assert_close(o2["out_position"], 2.0 * o1["out_position"], "skin linear in palette")
    assert np.isfinite(o1["joint_transforms"]).all() and np.isfinite(o1["out_position"]).all()


def test_c3_full_size_pose_and_skin_match_oracle(cuda_device):
    """BASELINE configs[2] at FULL size, the launch bench.py times: 50 000 characters x 64 joints, one distinct
    200-vertex mesh per character (10 M vertices, 50 000 distinct vert_first offsets), pose_update + skin against
    the oracle on EVERY output to 1e-5 -- the persistent grid wrapping 50 000 characters over the resident blocks,
    the ragged last block and the per-character vertex windows are all inside the comparison."""
    from clap_amd import animation
    J, n, vpc = 64, 50_000, 200
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    sk["bind"] = ob.skeleton_bind(sk)
    mesh = synth.skinned_mesh(vpc, J, seed=3, copies=n)                # what bench.py builds
    assert mesh["n_verts"] == n * vpc
    vf = (np.arange(n, dtype=np.int64) * vpc).astype(np.uint32)
    vc = np.full(n, vpc, np.uint32)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
    batch.set_frame_times(ch["phase"])
    batch.pose_update()
    batch.skin()
    out = batch.download()
    trs = np.tile(ch["trs0"], (n, 1, 1))
    jt, _g, jp = ob.pose(sk, an, ch["phase"], ch["char_mx"], trs)
    reach = sk["order"]
    assert_close(out["trs"], trs, "C3 full size T/R/S")
    assert_close(out["joint_transforms"][:, reach], jt[:, reach], "C3 full size joint_transforms")
    assert_close(out["joint_pos"][:, reach], jp[:, reach], "C3 full size joint pos")
    # per-character worst case too, so one bad character cannot hide in the global maximum
    d = np.abs(out["joint_transforms"][:, reach].astype(np.float64) - jt[:, reach]).reshape(n, -1).max(axis=1)
    s = np.abs(jt[:, reach]).reshape(n, -1).max(axis=1)
    assert float((d / s).max()) <= RTOL, f"worst character {int((d / s).argmax())}: {(d / s).max():.3e}"
    exp_p, exp_n = ob.skin(mesh, vf, vc, out["joint_transforms"])      # same palette in: isolates k_skin
    assert_close(out["out_position"], exp_p, "C3 full size skinned positions")
    assert_close(out["out_normal"], exp_n, "C3 full size skinned normals")
    exp_p2, exp_n2 = ob.skin(mesh, vf, vc, jt)                         # end to end against the oracle's palette
    assert_close(out["out_position"], exp_p2, "C3 full size pose -> skin positions")
    assert_close(out["out_normal"], exp_n2, "C3 full size pose -> skin normals")
    # per-vertex bound (positions scale with the character, so bound each vertex by its own character's extent)
    ext = np.abs(exp_p2).reshape(n, vpc * 3).max(axis=1)
    dv = np.abs(out["out_position"].astype(np.float64) - exp_p2).reshape(n, vpc * 3).max(axis=1)
    assert float((dv / np.maximum(ext, 1e-30)).max()) <= RTOL
