"""Per-character feeder in front of default_update (character.c:546-611).

CPU: the oracle against the reference's own character_update -> default_update for body-less
characters (golden fixture + live), and the body branch's invariants (parity unpinned: ODE).
GPU: the HIP kernel through the C ABI against the oracle, bit-exact (positions, history ring, flags,
body positions, moved)."""
import os

import numpy as np
import pytest

from clap_amd import synth
from oracle import binding as ob
from oracle import refrun

E_DIRTY = 1 << 16


def _chars(feed):
    return dict(entity=feed["entity"], body=feed["body"], hist_pos=feed["hist_pos"].copy(),
                hist_head=feed["hist_head"].copy(), hist_wrapped=feed["hist_wrapped"].copy(), airborne=feed["airborne"])


def _fall_frames(feed, frames=4):
    return np.stack([feed["pos"] + np.float32(f) * np.asarray([0, -30, 0], np.float32) for f in range(frames)])


def _run_oracle_frames(feed, pos_frames, scale):
    n = feed["n"]
    chars = _chars(feed)
    ps = np.zeros((n, 4), np.float32)
    ps[:, 3] = scale
    fl = np.zeros(n, np.uint32)
    out = dict(pos=[], hist_head=[], hist_wrapped=[])
    for f in range(len(pos_frames)):
        ps[:, :3] = pos_frames[f]
        ob.characters_update(chars, feed["limbo_height"], ps, fl)
        out["pos"].append(ps[:, :3].copy())
        out["hist_head"].append(chars["hist_head"].copy())
        out["hist_wrapped"].append(chars["hist_wrapped"].copy())
    return {k: np.stack(v) for k, v in out.items()}


# ---------------------------------------------------------------- CPU
def test_oracle_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "characters_limbo.npz"))
    feed = dict(n=len(z["in_hist_head"]), entity=np.arange(len(z["in_hist_head"]), dtype=np.uint32),
                body=np.full(len(z["in_hist_head"]), -1, np.int32), hist_pos=z["in_hist_pos"],
                hist_head=z["in_hist_head"], hist_wrapped=z["in_hist_wrapped"],
                airborne=np.zeros(len(z["in_hist_head"]), np.uint8), limbo_height=float(z["in_limbo_height"][0]))
    got = _run_oracle_frames(feed, z["in_pos_frames"], z["in_scale"])
    assert np.array_equal(got["pos"].view(np.uint32), z["ref_pos"].view(np.uint32)), "entity position after the hook"
    assert np.array_equal(got["hist_head"], z["ref_hist_head"]) and np.array_equal(got["hist_wrapped"], z["ref_hist_wrapped"])
    teleported = (got["pos"] != z["in_pos_frames"]).any(axis=2)
    assert teleported.any() and not teleported.all()
    # the chained default_update saw the teleported position: the reference's mx column 3 is that position
    assert np.array_equal(z["ref_mx"][..., 12:15].view(np.uint32), got["pos"].view(np.uint32))


@pytest.mark.skipif(not refrun.available(), reason="reference build (oracle/_ref) not present")
def test_oracle_matches_reference_live():
    n = 3000
    feed = synth.character_feed(n, seed=77, limbo_height=55.0, with_bodies=False)
    rng = np.random.Generator(np.random.PCG64(9))
    rot = synth.quat_from_euler_xyz(*rng.uniform(-3, 3, (3, n))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, n).astype(np.float32)
    frames = _fall_frames(feed, 5)
    ref = refrun.characters(frames, rot, scale, feed["hist_pos"], feed["hist_head"], feed["hist_wrapped"], 55.0)
    got = _run_oracle_frames(feed, frames, scale)
    for k in ("hist_head", "hist_wrapped"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(got["pos"].view(np.uint32), ref["pos"].view(np.uint32))


def test_oracle_body_branch_semantics():
    n = 500
    feed = synth.character_feed(n, seed=3, with_bodies=True)
    bodies = synth.sphere_bodies(n, box=50.0, seed=3)
    bodies["lvel"][::3] = 0.0                               # resting: no history_push, moved = 0
    bodies["pos"] = np.ascontiguousarray(feed["pos"].astype(np.float64))
    bodies["pos"][:, 1] += bodies["yoffset"]
    chars = _chars(feed)
    ps = np.zeros((n, 4), np.float32)
    ps[:, :3] = feed["pos"]
    fl = np.zeros(n, np.uint32)
    head0, wr0 = chars["hist_head"].copy(), chars["hist_wrapped"].copy()
    moved = ob.characters_update(chars, feed["limbo_height"], ps, fl, bodies)
    assert np.all(fl & E_DIRTY), "phys_body_update always repositions the entity"
    assert np.array_equal(moved != 0, np.linalg.norm(bodies["lvel"], axis=1) > 1e-3)
    assert np.array_equal(ps[:, 0], bodies["pos"][:, 0].astype(np.float32))
    assert np.array_equal(ps[:, 1], (bodies["pos"][:, 1] - bodies["yoffset"]).astype(np.float32))
    pushed = (moved != 0) & (feed["airborne"] == 0)
    # a teleport empties the ring (head = 0, wrapped cleared) before any push of the same frame
    reset = ~pushed & (head0 != 0) & (chars["hist_head"] == 0)
    assert reset.any() and not chars["hist_wrapped"][reset].any(), "some characters were teleported out of limbo"
    still = ~pushed & (chars["hist_head"] == head0) & (head0 != 0)
    assert np.array_equal(chars["hist_wrapped"][still], wr0[still])
    k = np.flatnonzero(pushed)[0]
    assert np.array_equal(chars["hist_pos"][k, (chars["hist_head"][k] - 1) % 8], ps[k, :3]), "newest entry = where it stands"


# ---------------------------------------------------------------- GPU
def _entity_scene(n, pos):
    scene = synth.pad_levels(synth.entities_flat(n, seed=1))
    scene["pos_scale"] = scene["pos_scale"].copy()
    scene["pos_scale"][:n, :3] = pos
    scene["flags"] = (scene["flags"] & ~np.uint32(E_DIRTY)).astype(np.uint32)
    return scene


@pytest.mark.gpu
@pytest.mark.parametrize("with_bodies", [False, True], ids=["no_bodies", "bodies"])
def test_hip_matches_oracle_over_frames(with_bodies, cuda_device):
    import torch
    from clap_amd import characters, entities, physics
    n = 5000
    feed = synth.character_feed(n, seed=21, with_bodies=with_bodies)
    scene = _entity_scene(n, feed["pos"])
    batch = entities.EntityBatch(scene, cuda_device)
    cf = characters.CharacterFeed(feed, cuda_device)
    world, bodies = None, None
    if with_bodies:
        bodies = synth.sphere_bodies(n, box=50.0, seed=21)
        bodies["lvel"][::4] = 0.0
        bodies["pos"] = np.ascontiguousarray(feed["pos"].astype(np.float64))
        bodies["pos"][:, 1] += bodies["yoffset"]
        world = physics.PhysWorld(bodies, None, device=cuda_device)
    chars = _chars(feed)
    ps = scene["pos_scale"].copy()
    fl = scene["flags"].copy()
    fr, _v, _p = entities.view_calc_frustum(synth.camera())
    for f in range(4):
        moved = ob.characters_update(chars, feed["limbo_height"], ps, fl, bodies)
        cf.character_update(batch, world)
        out = cf.download()
        assert np.array_equal(batch.pos_scale.cpu().numpy().view(np.uint32), ps.view(np.uint32)), f"frame {f} entity pos"
        assert np.array_equal(batch.flags.cpu().numpy().view(np.uint32), fl), f"frame {f} dirty flags"
        assert np.array_equal(out["hist_pos"].view(np.uint32), chars["hist_pos"].view(np.uint32))
        assert np.array_equal(out["hist_head"], chars["hist_head"]) and np.array_equal(out["hist_wrapped"], chars["hist_wrapped"])
        assert np.array_equal(out["moved"], moved)
        if with_bodies:
            assert np.array_equal(world.pos.cpu().numpy().view(np.uint64), bodies["pos"].view(np.uint64)), "body positions"
            # the bodies fall another 40 units: the next frame teleports more of them
            bodies["pos"][:, 1] -= 40.0
            world.pos.copy_(torch.from_numpy(bodies["pos"]))
        else:
            ps[:n, 1] -= np.float32(40.0)
            batch.pos_scale.copy_(torch.from_numpy(ps))
    if not with_bodies:
        assert ((fl & E_DIRTY) != 0).sum() > n // 10
        # the chained default_update rebuilds exactly the teleported entities
        batch.mq_update(fr)
        st = ob.entity_state(scene)
        st["flags"][:] = fl
        scene2 = dict(scene, pos_scale=ps)
        ob.entities_update(scene2, st)
        assert np.array_equal(batch.download()["mx"][((fl & E_DIRTY) != 0)].view(np.uint32),
                              st["mx"][((fl & E_DIRTY) != 0)].view(np.uint32))


@pytest.mark.gpu
def test_hip_c3_character_count_and_errors(cuda_device):
    import ctypes as C
    from clap_amd import _lib, characters, entities
    n = 50_000                                                    # BASELINE configs[2]: 50k characters
    feed = synth.character_feed(n, seed=5, with_bodies=False)
    scene = _entity_scene(n, feed["pos"])
    batch = entities.EntityBatch(scene, cuda_device)
    cf = characters.CharacterFeed(feed, cuda_device)
    cf.character_update(batch)
    chars = _chars(feed)
    ps, fl = scene["pos_scale"].copy(), scene["flags"].copy()
    ob.characters_update(chars, feed["limbo_height"], ps, fl)
    assert np.array_equal(batch.pos_scale.cpu().numpy().view(np.uint32), ps.view(np.uint32))
    assert np.array_equal(cf.download()["hist_head"], chars["hist_head"])
    desc = _lib.Characters(5, 70.0, None, None, None, None, None, None, None)
    assert _lib.lib().clapgpu_characters_update(None, C.byref(desc), C.byref(batch._desc), None) == _lib.ERR_INVALID_ARGUMENTS
    desc.n = 0
    assert _lib.lib().clapgpu_characters_update(None, C.byref(desc), C.byref(batch._desc), None) == _lib.OK
