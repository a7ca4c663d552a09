"""Joint attachments (parent_transform_apply's joint flavour, model.c:1626-1641) and the camera
bounding-volume pick (model.c:1703-1713): oracle vs the reference's golden vectors on CPU, HIP vs
both on GPU."""
import os

import numpy as np
import pytest

from clap_amd import synth, tiler
from oracle import binding as ob
from helpers import apply_frame, assert_bits_equal, load_golden

PATH = os.path.join(os.path.dirname(__file__), "golden", "attach_bv_frames.npz")


def load():
    scene, cam, ref, frames = load_golden(PATH)
    z = np.load(PATH)
    ent = z["in_attach_entity"]
    att = np.zeros(len(ent), ob.ATTACH_DTYPE)
    att["entity"], att["jt"], att["bind"] = ent, np.arange(len(ent)), np.arange(len(ent))
    return scene, cam, ref, frames, att, z["in_attach_jt"], z["in_attach_bind"], z["in_bv_cam_pos"], int(z["in_bv_ctl"][0])


def test_oracle_matches_reference_golden():
    scene, cam, ref, frames, att, jt, bind, cam_pos, ctl = load()
    st = ob.entity_state(scene)
    st["flags"] &= ~synth.E_DIRTY
    st["flags"][att["entity"]] |= np.uint32(1 << 17)
    assert ref["bv"].max() >= 0, "fixture must produce a bounding-volume pick"
    for f, frame in enumerate(frames):
        apply_frame(scene, st, frame)
        ob.entities_update_range(scene, st, 0, scene["n"], att, jt, bind)
        assert_bits_equal(st["mx"], ref["mx"][f], f"frame {f} mx")
        assert_bits_equal(st["inv_mx"], ref["inv_mx"][f], f"frame {f} inverse_mx")
        assert_bits_equal(st["aabb"], ref["aabb"][f], f"frame {f} aabb")
        assert np.array_equal(st["seqs"], ref["seqs"][f]), f"frame {f} seq / parent_seq"
        i, vol = ob.camera_bv(scene, st, cam_pos, scene["pos_scale"][ctl, :3], ctl)
        assert i == int(ref["bv"][f]) and np.float32(vol) == ref["bv_volume"][f], f"frame {f} bv pick"
    # attached entities are rebuilt every frame: their seq advances on all 3 frames
    alive = (scene["flags"][att["entity"]] & synth.E_ALIVE) != 0
    assert alive.sum() > 10 and np.all((ref["seqs"][2][att["entity"][alive]] & 0xFFFF) == 3)


@pytest.mark.gpu
def test_hip_matches_reference_golden(cuda_device):
    from clap_amd import entities
    scene, cam, ref, frames, att, jt, bind, cam_pos, ctl = load()
    fr, _v, _p = entities.view_calc_frustum(cam)
    scene["flags"] = scene["flags"] & ~synth.E_DIRTY
    batch = entities.EntityBatch(scene, cuda_device)
    batch.set_attachments(att, jt, bind)
    for f, (ps, rot, dirty) in enumerate(frames):
        idx = np.flatnonzero(dirty)
        batch.set_transforms(idx, ps[idx], rot[idx])
        batch.set_bv_query(cam_pos, ps[ctl, :3], ctl)
        batch.mq_update(fr)
        batch.compact_visible()
        out = batch.download()
        assert_bits_equal(out["mx"], ref["mx"][f], f"frame {f} mx")
        assert_bits_equal(out["inv_mx"], ref["inv_mx"][f], f"frame {f} inverse_mx")
        assert_bits_equal(out["aabb"], ref["aabb"][f], f"frame {f} aabb")
        assert np.array_equal(out["seqs"], ref["seqs"][f])
        assert np.array_equal(out["visible"], np.flatnonzero(ref["visible"][f]))
        i, vol = batch.camera_bv()
        assert i == int(ref["bv"][f]) and np.float32(vol) == ref["bv_volume"][f], f"frame {f} bv pick"


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["levels", "tiles"])
def test_hip_matches_oracle_with_pose_in_between(layout, cuda_device):
    """Frame order of the engine: characters' transforms -> pose -> the props riding their joints.
    jt_pool IS the pose kernel's joint_transforms output."""
    from clap_amd import animation, entities
    rng = np.random.Generator(np.random.PCG64(8))
    J, n_chars = 24, 40
    sk = synth.skeleton(J, 6, seed=8)
    an = synth.animation(J, 9, 2.0, seed=8)
    ch = synth.characters(n_chars, J, seed=8)
    sk["bind"] = ob.skeleton_bind(sk)
    base = synth.entities_forest(1500, seed=5, n_models=2, dead_frac=0.0)
    base["model_skip"][:] = 0
    scene = synth.pad_levels(base) if layout == "levels" else tiler.tiled_scene(base)[0]
    n = scene["n"]
    # characters = the first n_chars real root entities; props = some of their children
    roots = np.flatnonzero((scene["parent"] < 0) & (scene["orig_of"] >= 0))[:n_chars]
    kids = np.flatnonzero(np.isin(scene["parent"], roots))
    props = np.sort(rng.choice(kids, min(25, len(kids)), replace=False))
    char_of_root = {int(r): c for c, r in enumerate(roots)}
    att = np.zeros(len(props), ob.ATTACH_DTYPE)
    att["entity"] = props
    joints = rng.integers(0, J, len(props))
    att["jt"] = [char_of_root[int(scene["parent"][p])] * J + int(j) for p, j in zip(props, joints)]
    att["bind"] = joints
    cam = synth.camera(pos=(0, 10, 90))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)

    # ---- oracle: entities (props skipped by construction? no: props need this frame's palette) ----
    st = ob.entity_state(scene)
    st["flags"][props] |= np.uint32(1 << 17)
    # pass 1: everything except attached subtrees cannot be separated on the CPU side either; the
    # reference's converged order is: character mx -> its palette -> the prop.  Characters are roots
    # (level 0), so one full pass with a palette computed from the characters' matrices is converged.
    tmp = {k: v.copy() for k, v in st.items()}
    ob.entities_update_range(scene, tmp, 0, n)                         # character matrices only matter here
    trs = np.tile(ch["trs0"], (n_chars, 1, 1))
    jt, _g, _pos = ob.pose(sk, an, ch["phase"][:n_chars], tmp["mx"][roots], trs)
    ob.entities_update_range(scene, st, 0, n, att, jt.reshape(-1, 16), sk["bind"])
    vis, mask = ob.entities_cull(n, st["flags"], st["aabb"], fr_o)

    # ---- GPU: update -> pose (reads entity mx on the device) -> update again for the props ----
    batch = entities.EntityBatch(scene, cuda_device)
    model = animation.SkinnedModel(sk, [an], bind=sk["bind"], device=cuda_device)
    cb = animation.CharacterBatch(model, n_chars, ch["trs0"], batch.mx, entity_index=roots.astype(np.uint32))
    cb.set_frame_times(ch["phase"][:n_chars])
    batch.set_attachments(att, cb.joint_transforms.view(-1, 16), sk["bind"])
    batch.mq_update(fr)                  # props use a stale palette here ...
    cb.pose_update()
    batch.mq_update(fr)                  # ... and are rebuilt (always) with this frame's palette
    batch.compact_visible()
    out = batch.download()
    assert_bits_equal(out["mx"], st["mx"], "riders of a GPU palette: bit-exact since round 4 (the pose is the reference's arithmetic)")
    nonprop = np.ones(n, bool)
    desc = props.copy()
    for _ in range(8):                   # props' subtrees depend on the palette (bit-exact since round 4)
        desc = np.union1d(desc, np.flatnonzero(np.isin(scene["parent"], desc)))
    nonprop[desc] = False
    assert_bits_equal(out["mx"][nonprop], st["mx"][nonprop], "entities not under a prop stay bit-exact")
    assert int(out["visible_count"]) == len(vis) and np.array_equal(out["visible"], vis)
