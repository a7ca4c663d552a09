"""GPU: pose blend + joint palette + vertex skinning (through the C ABI) against the oracle and
the reference's golden vectors.

BASELINE.json north_star asks for "within 1e-5 relative for float transforms/positions".  Since round 4 the kernel performs
the reference's operations in the reference's order and roundings (clap_amd/csrc/pose.hip: IEEE quotients, fp64 lerp,
the slerp's acos / sin of each key pair from the host's libm at model build, fp64 sin / cos of the frame's angle, the
hierarchy level by level in the association of model.c:1363-1383), so every comparison here is EQUALITY of bit patterns:
each joint's T, R and S, its palette matrix, its world position, each skinned vertex position and normal -- against the
oracle and against the golden vectors of the reference itself (tests/helpers.py assert_values_equal; what was compared
goes to gpurun_out/parity_bounds.json).
"""
import glob
import os

import numpy as np
import torch
import pytest

from clap_amd import synth
from oracle import binding as ob
from helpers import assert_mat4_equal, assert_trs_equal, assert_vec_equal
from test_oracle_pose import load_pose

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pose_*.npz")))


def assert_pose_equal(out, trs, jt, jp, reach, what, trs_on=True, pos_on=True):
    """Every joint's T, R, S, its palette matrix and its world position: the reference's values."""
    if trs_on:
        assert_trs_equal(out["trs"], trs, what + " T/R/S", key="pose T/R/S")
    assert_mat4_equal(out["joint_transforms"][:, reach], jt[:, reach], what + " joint_transforms", key="pose joint_transforms")
    if pos_on:
        assert_vec_equal(out["joint_pos"][:, reach], jp[:, reach], what + " joint pos", key="pose joint_pos")


def oracle_pose(sk, anims, which, t, ent_mx, trs):
    """The oracle for a batch whose characters play different animations: (joint_transforms, globals, joint_pos);
    trs is updated in place."""
    n, J = trs.shape[0], sk["nr_joints"]
    jt = np.zeros((n, J, 16), np.float32)
    gl = np.zeros((n, J, 16), np.float32)
    jp = np.zeros((n, J, 4), np.float32)
    for a_id, an in enumerate(anims):
        sel = np.flatnonzero(which == a_id)
        if not sel.size:
            continue
        sub = trs[sel].copy()
        j1, g1, p1 = ob.pose(sk, an, t[sel], ent_mx[sel], sub)
        trs[sel], jt[sel], gl[sel], jp[sel] = sub, j1, g1, p1
    return jt, gl, jp


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
        assert_pose_equal(out, ref["trs"][f], ref["joint_transforms"][f], ref["joint_pos"][f], reach, f"frame {f}")
        assert not out["joint_transforms"][:, unreach].any(), "joints outside joint 0's tree must stay untouched"


@pytest.mark.parametrize("J,depth,skw,akw", [
    (64, 8, {}, {}),                                                   # BASELINE config 3 skeleton
    (64, 8, dict(unreachable=6), dict(missing_frac=0.25, ragged=True)),
    (5, 3, {}, {}),
    (100, 12, {}, dict(ragged=True)),                                  # two wavefronts per character
    (200, 20, dict(unreachable=3), dict(missing_frac=0.1)),            # JOINTS_MAX, general loop (missing channels)
    (128, 10, {}, {}),                                                 # two full wavefronts per character, streaming loop
    (192, 14, {}, dict(ragged=True)),                                  # three (192-thread workgroups)
    (200, 16, {}, {}),                                                 # four, the last one 8 joints wide
], ids=["c3_64j", "ragged_64j", "tiny_5j", "two_wave_100j", "joints_max_200j", "stream_128j", "stream_192j", "stream_200j"])
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
        jt, gl, jp = oracle_pose(sk, (an0, an1), which, t, ch["char_mx"], trs)
        batch.set_frame_times(t, which)
        batch.pose_update()
        out = batch.download()
        assert_pose_equal(out, trs, jt, jp, reach, f"frame {f}")


@pytest.mark.parametrize("J", [64, 40, 1, 100, 200], ids=["64j", "40j_short_rows", "one_joint", "100j_two_waves", "200j_four_waves"])
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
    jt, gl, jp = oracle_pose(sk, (an0, an1), which, t, ent_mx[perm], trs)
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
        assert_pose_equal(out, trs, jt, jp, reach, what, trs_on=trs_on, pos_on=pos_on)
        if not trs_on:
            assert (out["trs"] == 7.5).all(), what + ": masked T/R/S was written"
        if not pos_on:
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
        jt, gl, jp = oracle_pose(sk, anims, which, t, ch["char_mx"], trs)
        batch.set_frame_times(t, which)
        batch.pose_update()
        out = batch.download()
        assert_pose_equal(out, trs, jt, jp, reach, f"frame {f}")


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
        assert_mat4_equal(batch.download()["joint_transforms"][:, reach], z["ref_joint_transforms"][f][:, reach], f"frame {f}",
                          key="pose joint_transforms")
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
    assert_vec_equal(out["out_position"], exp_p, "skinned positions", key="skin position (same palette)")
    assert_vec_equal(out["out_normal"], exp_n, "skinned normals", key="skin normal (same palette)")
    # and end to end against the oracle's own palette
    trs = np.tile(ch["trs0"], (n, 1, 1))
    jt, _gl, _p = ob.pose(sk, an, ch["phase"], ch["char_mx"], trs)
    exp_p2, exp_n2 = ob.skin(mesh, vf, vc, jt)
    assert_vec_equal(out["out_position"], exp_p2, "pose -> skin end to end", key="pose -> skin position")
    assert_vec_equal(out["out_normal"], exp_n2, "pose -> skin end to end (normals)", key="pose -> skin normal")


def test_skin_w_output_with_unnormalised_weights(cuda_device):
    """shaders/model.vert:32-45: the shader feeds proj * view * trs the VEC4 total_local_pos, whose w is the sum of the
    vertex's weights (times row 3 of the palette matrices) -- never renormalised.  With clapgpu_skin_batch.out_w the
    kernel emits that w; vec4(out_position, out_w) is then what the shader would have used, also for weights that
    do not sum to 1.  Same palette in on both sides: bit-exact against oracle/skin.c's fourth component."""
    from clap_amd import animation
    from helpers import assert_bits_equal
    J, n = 64, 19
    sk = synth.skeleton(J, 8, seed=3)
    an = synth.animation(J, 30, 2.0, seed=3)
    ch = synth.characters(n, J, seed=3)
    sk["bind"] = ob.skeleton_bind(sk)
    vc = np.random.Generator(np.random.PCG64(9)).integers(1, 600, n).astype(np.uint32)
    mesh = synth.skinned_mesh(int(vc.sum()), J, seed=5)
    rng = np.random.Generator(np.random.PCG64(10))
    mesh["weights"] = (mesh["weights"] * rng.uniform(0.25, 1.75, (mesh["n_verts"], 1))).astype(np.float32)   # sums in [0.25, 1.75]
    mesh["weights"][::7] = 0.0                                          # and vertices without any influence: w = 0
    vf = np.concatenate([[0], np.cumsum(vc[:-1])]).astype(np.uint32)
    model = animation.SkinnedModel(sk, [an], mesh=mesh, bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"], vert_first=vf, vert_count=vc)
    batch.set_frame_times(ch["phase"])
    batch.pose_update()
    batch.skin()
    o0 = batch.download()
    assert "out_w" not in o0
    batch.set_skin_w(True)
    batch.out_w.fill_(-7.0)
    batch.skin()
    o1 = batch.download()
    ep, en, ew = ob.skin(mesh, vf, vc, o1["joint_transforms"], with_w=True)
    assert_bits_equal(o1["out_position"], ep, "positions (w on)")
    assert_bits_equal(o1["out_normal"], en, "normals (w on)")
    assert_bits_equal(o1["out_w"], ew, "total_local_pos.w")
    assert_bits_equal(o1["out_position"], o0["out_position"], "positions do not depend on the w output")
    wsum = mesh["weights"].astype(np.float64).sum(axis=1)
    cat = np.concatenate([np.arange(f, f + k) for f, k in zip(vf, vc)])
    assert np.abs(o1["out_w"] - wsum[cat]).max() <= 1e-6, "affine palettes: w is the sum of the weights"
    assert np.abs(o1["out_w"] - 1.0).max() > 0.5, "the fixture's weights are far from normalised"
    batch.set_skin_w(False)                                            # and off again: the buffer is left alone
    batch.out_w.fill_(-7.0)
    batch.skin()
    assert (batch.out_w == -7.0).all()


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
    assert_vec_equal(o2["out_position"], 2.0 * o1["out_position"], "skin linear in palette")
    assert np.isfinite(o1["joint_transforms"]).all() and np.isfinite(o1["out_position"]).all()


def test_c3_full_size_pose_and_skin_match_oracle(cuda_device):
    """BASELINE configs[2] at FULL size, the launch bench.py times: 50 000 characters x 64 joints, one distinct
    200-vertex mesh per character (10 M vertices, 50 000 distinct vert_first offsets), pose_update + skin against
    the oracle on EVERY output, bit for bit -- the persistent grid wrapping 50 000 characters over the resident blocks,
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
    jt, gl, jp = ob.pose(sk, an, ch["phase"], ch["char_mx"], trs)
    reach = sk["order"]
    assert_pose_equal(out, trs, jt, jp, reach, "C3 full size")
    exp_p, exp_n = ob.skin(mesh, vf, vc, jt)                           # end to end: the oracle's pose -> the oracle's skinning
    assert_vec_equal(out["out_position"], exp_p, "C3 full size pose -> skin positions", key="pose -> skin position")
    assert_vec_equal(out["out_normal"], exp_n, "C3 full size pose -> skin normals", key="pose -> skin normal")


def test_pose_stops_at_model_space_and_joint_pos_world_finishes(cuda_device):
    """CLAPGPU_POSE_JOINT_POS_MODEL (what the overlapped frame uses: the pose no longer waits for the frame's entity
    update): joint_pos holds column 3 of joint_transforms * bind (model.c:1392-1397), the entity matrices are not read
    (poisoned here), and clapgpu_joint_pos_world() then gives the bits of the fused path (model.c:1400)."""
    import torch
    from clap_amd import animation
    J, n = 100, 777                                                   # two wavefronts per character, an entity index list
    sk = synth.skeleton(J, 9, seed=41, unreachable=3)
    an = synth.animation(J, 12, 1.5, seed=41)
    ch = synth.characters(n, J, seed=41)
    sk["bind"] = ob.skeleton_bind(sk)
    rng = np.random.default_rng(8)
    perm = rng.permutation(n + 5).astype(np.uint32)[:n]
    ent_mx = rng.standard_normal((n + 5, 16)).astype(np.float32)
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ent_mx, entity_index=perm)
    t = ch["phase"].astype(np.float32)
    batch.set_frame_times(t)
    batch.pose_update()
    fused = batch.download()
    trs = np.tile(ch["trs0"], (n, 1, 1))
    jt, _gl, jp = ob.pose(sk, an, t, ent_mx[perm], trs)
    reach = sk["order"]
    assert_pose_equal(fused, trs, jt, jp, reach, "fused")
    batch.joint_pos.fill_(-5.0)
    batch.set_joint_pos_model_space(True)
    good = batch.entity_mx.clone()
    batch.entity_mx.fill_(float("nan"))                               # must not be read
    batch.pose_update()
    batch.entity_mx.copy_(good)
    model_space = batch.download()
    assert np.array_equal(model_space["joint_transforms"], fused["joint_transforms"])
    ident = np.tile(np.eye(4, dtype=np.float32).reshape(16), (n, 1))
    _jt, _g, mp = ob.pose(sk, an, t, ident, np.tile(ch["trs0"], (n, 1, 1)))       # identity entity: pos == mpos
    assert np.array_equal(model_space["joint_pos"][:, reach], mp[:, reach])
    assert (model_space["joint_pos"][:, np.setdiff1d(np.arange(J), reach)] == -5.0).all()
    batch.joint_pos_world()
    torch.cuda.synchronize()
    done = batch.download()["joint_pos"]
    assert np.array_equal(done[:, reach], fused["joint_pos"][:, reach])
    assert (done[:, np.setdiff1d(np.arange(J), reach)] == -5.0).all(), "joints outside joint 0's tree stay untouched"


def test_pose_update_rejects_pools_made_for_something_else(cuda_device):
    """clapgpu_pose_update needs the pools of clapgpu_animations_pack for THIS skeleton class and animation count (they hold
    what the reference's slerp derives from each key pair with the host's libm): none, another joint class, another
    animation count, a stale layout word -> CLAPGPU_ERR_INVALID_ARGUMENTS, nothing launched."""
    import ctypes as C
    from clap_amd import _lib, animation
    sk64, sk100 = synth.skeleton(64, 6, seed=2), synth.skeleton(100, 6, seed=2)
    an64, an100 = synth.animation(64, 8, 1.0, seed=2), synth.animation(100, 8, 1.0, seed=2)
    m64 = animation.SkinnedModel(sk64, [an64], device=cuda_device)
    m100 = animation.SkinnedModel(sk100, [an100], device=cuda_device)
    ch = synth.characters(5, 64, seed=2)
    batch = animation.CharacterBatch(m64, 5, ch["trs0"], ch["char_mx"])
    batch.set_frame_times(ch["phase"])
    L = _lib.lib()
    call = lambda desc: L.clapgpu_pose_update(None, C.byref(m64.skel_desc), C.byref(desc), C.byref(batch._pose_desc))
    assert call(m64.anim_desc) == 0
    keep = (m64.anim_desc.packed, m64.anim_desc.packed_keys, m64.anim_desc.packed_layout, m64.anim_desc.n_anims)
    m64.anim_desc.packed = None
    assert call(m64.anim_desc) == _lib.ERR_INVALID_ARGUMENTS, "no pools"
    m64.anim_desc.packed = keep[0]
    m64.anim_desc.packed_layout = m100.anim_desc.packed_layout            # pools of a two-wavefront skeleton
    assert call(m64.anim_desc) == _lib.ERR_INVALID_ARGUMENTS
    m64.anim_desc.packed_layout = 0                                       # never packed
    assert call(m64.anim_desc) == _lib.ERR_INVALID_ARGUMENTS
    m64.anim_desc.packed_layout = keep[2]
    m64.anim_desc.n_anims = 2                                             # another animation count than the pools were made for
    assert call(m64.anim_desc) == _lib.ERR_INVALID_ARGUMENTS
    m64.anim_desc.n_anims = keep[3]
    assert call(m64.anim_desc) == 0


def test_pack_refuses_key_times_that_are_not_strictly_increasing(cuda_device):
    """The kernel's branch-free bracket (the number of keys below the time) equals channel_time_to_idx
    (core/model.c:1266-1288) for strictly increasing key times only; the reference's cursor-dependent scan gives other
    pairs on equal or descending times.  clapgpu_animations_pack has every channel on the host once per model: it
    refuses such an asset (so that the caller keeps the model on the host path) instead of mis-posing it every frame;
    likewise a channel record that reaches past its pool."""
    from clap_amd import _lib, animation
    sk = synth.skeleton(64, 6, seed=5)
    good = synth.animation(64, 8, 1.0, seed=5)
    animation.SkinnedModel(sk, [good], device=cuda_device)                  # packs
    for kind in ("equal", "descending"):
        bad = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in good.items()}
        c = 40                                                              # some channel in the middle
        off, nr = int(bad["ch_time_off"][c]), int(bad["ch_nr"][c])
        assert nr >= 4
        if kind == "equal":
            bad["times"][off + 2] = bad["times"][off + 1]
        else:
            bad["times"][off + 1], bad["times"][off + 2] = bad["times"][off + 2], bad["times"][off + 1]
        with pytest.raises(Exception) as ei:
            animation.SkinnedModel(sk, [bad], device=cuda_device)
        assert "clapgpu_animations_pack" in str(ei.value)
        assert "strictly increasing" in _lib.lib().clapgpu_last_error().decode()
    past = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in good.items()}
    past["ch_data_off"][-1] += 8                                            # the last channel's values end past data[]
    with pytest.raises(Exception):
        animation.SkinnedModel(sk, [past], device=cuda_device)


def test_palette_only_mode_is_refused_where_a_path_has_no_channel(cuda_device):
    """A (joint, path) without a channel keeps the joint's LAST interpolated value (core/model.c:1301), which lives in
    trs[]: with CLAPGPU_POSE_SKIP_TRS trs[] is never written, and a character that switches from an animation with the
    channel to one without it would pose from its first frame's values.  The combination is refused."""
    import ctypes as C
    from clap_amd import _lib, animation
    sk = synth.skeleton(64, 6, seed=6)
    full, holes = synth.animation(64, 8, 1.0, seed=6), synth.animation(64, 8, 1.0, seed=7, missing_frac=0.2)
    ch = synth.characters(4, 64, seed=6)
    L = _lib.lib()
    for anims, ok in (([full], True), ([full, holes], False)):
        m = animation.SkinnedModel(sk, anims, device=cuda_device)
        b = animation.CharacterBatch(m, 4, ch["trs0"], ch["char_mx"])
        b.set_frame_times(ch["phase"])
        b.set_outputs(trs=False, joint_pos=True)
        rc = L.clapgpu_pose_update(None, C.byref(m.skel_desc), C.byref(m.anim_desc), C.byref(b._pose_desc))
        assert rc == (0 if ok else _lib.ERR_INVALID_ARGUMENTS)
        if not ok:
            assert "no channel" in L.clapgpu_last_error().decode()
            b.set_outputs(trs=True, joint_pos=True)
            assert L.clapgpu_pose_update(None, C.byref(m.skel_desc), C.byref(m.anim_desc), C.byref(b._pose_desc)) == 0


def _skeleton_with_parents(parent, seed):
    """A synth.skeleton() whose tree is replaced by `parent` (joint 0 the root; -2 = not under joint 0)."""
    J = len(parent)
    sk = synth.skeleton(J, 2, seed=seed)
    parent = np.asarray(parent, np.int32)
    depth = np.full(J, -1, np.int32)
    depth[0] = 0
    order = [0]
    for _ in range(J):
        for j in range(1, J):
            if depth[j] < 0 and parent[j] >= 0 and depth[parent[j]] >= 0:
                depth[j] = depth[parent[j]] + 1
    reach = [j for j in np.argsort(depth, kind="stable") if depth[j] >= 0]
    sk["parent"] = np.where(parent == -2, -1, parent).astype(np.int32)
    sk["depth"] = depth
    sk["order"] = np.asarray(reach, np.int32)
    return sk


@pytest.mark.parametrize("shape", ["chain_64", "star_64", "chain_200", "comb_150", "two_roots_40", "understated_levels"])
def test_pose_level_passes_on_odd_trees(shape, cuda_device):
    """The hierarchy runs as level passes of up to LPC / 4 joints scheduled on the device from parent[] / depth[] (joints with
    children first, left-over slots topped up from the next level), with a per-lane program in dynamic LDS sized by the
    host from n_levels: a 64-level chain (64 passes of one joint), a star (63 siblings: four passes of one level), a
    200-level chain (more passes than the LDS left beside the key times holds: the times go through L2; what still does
    not fit is computed on the fly), a comb (a spine with a tooth per vertebra: every pass mixes two levels), a second
    root whose subtree joint 0 does not hold, and a caller that understates n_levels -- all EQUAL to the oracle."""
    from clap_amd import animation
    if shape == "chain_64":
        parent = [-1] + list(range(63))
    elif shape == "star_64":
        parent = [-1] + [0] * 63
    elif shape == "chain_200":
        parent = [-1] + list(range(199))
    elif shape == "comb_150":
        parent = [-1] + [(j - 1) if j % 2 else (j - 2) for j in range(1, 150)]       # odd joints: the spine's teeth
        parent = [-1] + [max(p, 0) for p in parent[1:]]
    elif shape == "two_roots_40":
        parent = [-1] + [int(j // 2) for j in range(1, 30)] + [-2] + list(range(30, 39))
    else:
        parent = [-1] + [int((j - 1) // 3) for j in range(1, 90)]
    sk = _skeleton_with_parents(parent, seed=61)
    J, n = sk["nr_joints"], 41
    an = synth.animation(J, 9, 1.5, seed=61)
    ch = synth.characters(n, J, seed=61)
    sk["bind"] = ob.skeleton_bind(sk)
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    if shape == "understated_levels":
        model.skel_desc.n_levels = 1                                     # the program table is then too small: the rest on the fly
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    trs = np.tile(ch["trs0"], (n, 1, 1))
    reach = sk["order"]
    for f, t in enumerate((ch["phase"], (ch["phase"] * 0.7 + 0.31) % 1.6)):
        t = t.astype(np.float32)
        jt, _gl, jp = ob.pose(sk, an, t, ch["char_mx"], trs)
        batch.set_frame_times(t)
        batch.pose_update()
        out = batch.download()
        assert_pose_equal(out, trs, jt, jp, reach, f"{shape} frame {f}")
    unreach = np.setdiff1d(np.arange(J), reach)
    assert not out["joint_transforms"][:, unreach].any(), "joints outside joint 0's tree stay untouched"
    if shape == "two_roots_40":
        assert len(unreach) == 10


def test_pose_signed_zeros_are_the_references(cuda_device):
    """mat4x4_mul starts every sum from +0 (linmath.h:506-516 "t = 0.f; t += ..."), so a sum of zero products is +0 there
    whatever their signs, while mat4x4_scale_aniso and mat4x4_mul_vec4_post keep a -0.  A model made of exact zeros,
    ones and negative numbers -- identity and half-turn rotations, zero and negative translations, mirrored scales,
    axis-aligned bind and entity matrices -- makes most products a signed zero: the same BITS as the oracle (the palette
    holds thousands of zeros and not one -0; a kernel that sums from the first product would have returned hundreds)."""
    from clap_amd import animation
    rng = np.random.default_rng(77)
    J, n = 48, 37
    sk = synth.skeleton(J, 7, seed=77)
    quats = np.array([[0, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, -1], [-1, 0, 0, 0]], np.float32)

    def axis_mats(k, tr):
        m = np.zeros((k, 4, 4), np.float32)                             # [col][row]: a signed permutation + a translation
        for i in range(k):
            perm = rng.permutation(3)
            for c in range(3):
                m[i, c, perm[c]] = rng.choice([-1.0, 1.0, 2.0, -0.5])
            m[i, 3, :3] = rng.choice(tr, 3)
            m[i, 3, 3] = 1.0
        return m.reshape(k, 16)

    sk["invmx"] = axis_mats(J, [0.0, -0.0, 1.0, -2.0])
    sk["root_pose"] = axis_mats(1, [0.0, -1.0])[0]
    sk["bind"] = ob.skeleton_bind(sk)
    an = synth.animation(J, 5, 1.0, seed=77, missing_frac=0.2)
    data, off = an["data"].copy(), 0
    for c in range(an["n_channels"]):                                   # every key an exact value: lerps of equal ends, slerps of axis quaternions
        k, p, o = int(an["ch_nr"][c]), int(an["ch_path"][c]), int(an["ch_data_off"][c])
        if p == 1:
            data[o:o + 4 * k] = quats[rng.integers(0, len(quats), k)].ravel()
        elif p == 0:
            data[o:o + 3 * k] = np.tile(rng.choice([0.0, -0.0, -1.0, 0.5], 3), k)
        else:
            data[o:o + 3 * k] = np.tile(rng.choice([1.0, -1.0, 2.0], 3), k)
    an["data"] = data.astype(np.float32)
    ch = synth.characters(n, J, seed=77)
    ch["char_mx"] = axis_mats(n, [0.0, -0.0, 3.0, -4.0])
    ch["trs0"][:, 0:3] = rng.choice([0.0, -0.0, 1.0], (J, 3))
    ch["trs0"][:, 3:7] = quats[rng.integers(0, len(quats), J)]
    ch["trs0"][:, 7:10] = rng.choice([1.0, -1.0], (J, 3))
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ch["char_mx"])
    trs = np.tile(ch["trs0"], (n, 1, 1))
    t = ch["phase"].astype(np.float32) % 1.0
    t[:5] = [0.0, 1.0, 0.5, 0.25, 0.75]
    jt, _gl, jp = ob.pose(sk, an, t, ch["char_mx"], trs)
    batch.set_frame_times(t)
    batch.pose_update()
    out = batch.download()
    neg0 = lambda a: int(((a == 0) & np.signbit(a)).sum())
    assert neg0(trs) > 1000 and neg0(jt) == 0 and ((jt == 0) & ~np.signbit(jt)).sum() > 5000, \
        "the case is meant to hold -0 inputs, and mat4x4_mul never returns one"
    assert_pose_equal(out, trs, jt, jp, sk["order"], "signed zeros")


def test_pose_blocks_that_take_hundreds_of_characters(cuda_device, monkeypatch):
    """A persistent block keeps the per-character scalars (animation, frame time, entity) of its next 64 characters in the
    lanes of three registers and reloads them every 64 iterations; at BASELINE sizes a block sees 17 characters per slot.
    Two blocks for 3 000 characters (CLAPGPU_POSE_BLOCKS) make 125 iterations each: two animations, per-character
    entity indices, bit for bit."""
    from clap_amd import animation
    monkeypatch.setenv("CLAPGPU_POSE_BLOCKS", "2")
    J, n = 64, 3000
    sk = synth.skeleton(J, 8, seed=5)
    anims = [synth.animation(J, 12, 2.0, seed=5), synth.animation(J, 7, 1.1, seed=6, ragged=True)]
    ch = synth.characters(n, J, seed=5)
    sk["bind"] = ob.skeleton_bind(sk)
    rng = np.random.default_rng(5)
    which = rng.integers(0, 2, n).astype(np.uint32)
    ent = rng.permutation(n + 50)[:n].astype(np.uint32)                 # the characters' entities: any rows of a larger table
    ent_mx = np.zeros((n + 50, 16), np.float32)
    ent_mx[ent] = ch["char_mx"]
    model = animation.SkinnedModel(sk, anims, bind=sk["bind"], device=cuda_device)
    batch = animation.CharacterBatch(model, n, ch["trs0"], ent_mx, entity_index=ent)
    batch.anim.copy_(torch.from_numpy(which.astype(np.int32)).to(batch.anim.dtype))
    trs = np.tile(ch["trs0"], (n, 1, 1))
    t = ch["phase"].astype(np.float32)
    jt, _gl, jp = oracle_pose(sk, anims, which, t, ch["char_mx"], trs)
    batch.set_frame_times(t)
    batch.pose_update()
    out = batch.download()
    assert_pose_equal(out, trs, jt, jp, sk["order"], "125 characters per block slot")
