"""GPU: the HIP entity path (through the C ABI) against the oracle and the golden vectors.

Bar: bit-exact mx / inverse_mx / aabb / aabb_center / seq state / visibility mask /
ascending visible list."""
import glob
import os

import numpy as np
import pytest

from clap_amd import synth, tiler
from oracle import binding as ob
from helpers import apply_frame, assert_bits_equal, load_golden

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "entities_*.npz")))


LAYOUTS = ["levels", "tiles"]


def lay_out(scene, layout):
    """levels: level-major, one launch per level; tiles: subtree tiles, one launch in all."""
    return synth.pad_levels(scene) if layout == "levels" else tiler.tiled_scene(scene)[0]


def oracle_frame(scene, st, fr_o):
    ob.entities_update(scene, st)
    return ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr_o)


def check_against(out, st, vis, mask, what, check_flags=True):
    assert_bits_equal(out["mx"], st["mx"], what + " mx")
    assert_bits_equal(out["inv_mx"], st["inv_mx"], what + " inverse_mx")
    assert_bits_equal(out["aabb"], st["aabb"], what + " aabb")
    assert_bits_equal(out["center"], st["center"], what + " aabb_center")
    assert np.array_equal(out["seqs"], st["seqs"]), what + " seqs"
    if check_flags:
        assert np.array_equal(out["flags"], st["flags"]), what + " flags"
    assert np.array_equal(out["vis_mask"][:mask.size], mask), what + " vis_mask"
    assert out["visible_count"] == vis.size, what + " visible count"
    assert np.array_equal(out["visible"], vis), what + " visible list"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_reference_golden(path, cuda_device):
    from clap_amd import entities
    scene, cam, ref, frames = load_golden(path)
    n = scene["n"]
    fr, _view, _proj = entities.view_calc_frustum(cam)
    scene["flags"] = scene["flags"] & ~synth.E_DIRTY
    batch = entities.EntityBatch(scene, cuda_device)
    if frames is None:
        frames = [(scene["pos_scale"], scene["rot"], np.ones(n, np.uint8))]
    for f, (ps, rot, dirty) in enumerate(frames):
        idx = np.flatnonzero(dirty)
        batch.set_transforms(idx, ps[idx], rot[idx])
        batch.mq_update(fr)
        batch.compact_visible()
        out = batch.download()
        assert_bits_equal(out["mx"], ref["mx"][f], f"frame {f} mx")
        assert_bits_equal(out["inv_mx"], ref["inv_mx"][f], f"frame {f} inverse_mx")
        assert_bits_equal(out["aabb"], ref["aabb"][f], f"frame {f} aabb")
        assert_bits_equal(out["center"], ref["center"][f], f"frame {f} center")
        assert np.array_equal(out["seqs"], ref["seqs"][f]), f"frame {f} seqs"
        assert np.array_equal(out["visible"], np.flatnonzero(ref["visible"][f])), f"frame {f} visible list"


@pytest.mark.parametrize("maker,camkw", [
    (lambda: synth.entities_flat(10_000, 1234), {}),                       # BASELINE config 1
    (lambda: synth.entities_flat(10_000, 77, True), dict(ndc_z_zero_one=1)),
    (lambda: synth.entities_forest(20_000, 3, max_depth=9, n_models=5), dict(pos=(0, 20, 100))),
    (lambda: synth.entities_chains(3_000, 8, 2), {}),
    (lambda: synth.entities_flat(1, 5), {}),                               # single entity
    (lambda: synth.entities_flat(63, 6), {}),                              # less than one wave
    (lambda: synth.entities_flat(65, 7), {}),                              # one wave + 1
    (lambda: synth.entities_chains(1, 12, 9), {}),                         # one 12-deep chain, 64-padded levels
], ids=["c1_flat_10k", "flat_euler_z01", "forest_20k", "chains_3k_x8", "n1", "n63", "n65", "deep_chain"])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_hip_matches_oracle(maker, camkw, layout, cuda_device):
    from clap_amd import entities
    scene = lay_out(maker(), layout)
    cam = synth.camera(**camkw)
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    vis, mask = oracle_frame(scene, st, fr_o)

    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr)
    batch.compact_visible()
    check_against(batch.download(), st, vis, mask, "fused update+cull")
    batch.visible.zero_()
    batch.compact_visible(index_base=7, two_pass=True)       # large-n compaction path, shard offset
    out = batch.download()
    assert np.array_equal(out["visible"], vis + 7) and out["visible_count"] == vis.size

    # second frame with nothing dirty: nothing may change (seq/parent_seq skip path), cull repeats
    vis2, mask2 = oracle_frame(scene, st, fr_o)
    batch.mq_update(fr)
    batch.compact_visible()
    check_against(batch.download(), st, vis2, mask2, "clean second frame")
    assert np.array_equal(vis, vis2)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_all_dirty_mode_and_separate_cull(layout, cuda_device):
    """CLAPGPU_UPDATE_ALL_DIRTY (what bench.py times) == every entity marked dirty; the
    stand-alone cull pass == the fused one."""
    from clap_amd import entities
    scene = lay_out(synth.entities_forest(8_000, 31), layout)
    cam = synth.camera(pos=(3, 4, 60))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    for frame in range(3):
        st["flags"] |= np.where(st["flags"] & synth.E_ALIVE, synth.E_DIRTY, 0).astype(np.uint32)
        vis, mask = oracle_frame(scene, st, fr_o)
    batch = entities.EntityBatch(scene, cuda_device)
    for frame in range(3):
        batch.mq_update(None, all_dirty=True)
    batch.cull(fr)
    batch.compact_visible()
    out = batch.download()
    check_against(out, st, vis, mask, "all-dirty x3 + cull", check_flags=False)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    out2 = batch.download()
    assert np.array_equal(out2["visible"], out["visible"])


@pytest.mark.parametrize("n_models,dead", [(300, 0.0), (1, 0.0), (40, 0.002), (256, 0.0)],
                         ids=["300_models_no_lds_table", "one_model_all_rows_straight", "rare_dead_rows_leave_the_loop",
                              "256_models_table_full"])
def test_tile_kernel_row_loops(n_models, dead, cuda_device):
    """The tile kernel runs rows whose 64 lanes are all alive, rebuilt and boxed through a straight-line loop with
    the model table in LDS (<= 256 models) and leaves it for the general loop at the first row that does not
    qualify: a table too large for LDS (general loop only), a scene where every row qualifies, one where a few dead
    entities / models without a box (the last model skips its AABB) break rows in the middle of tiles, and the
    largest cached table -- three frames each, all dirty and then a tenth dirty, bit-exact against the oracle."""
    from clap_amd import entities
    rng = np.random.Generator(np.random.PCG64(n_models))
    raw = synth.entities_forest(20_000, 40 + n_models, max_depth=8, n_models=n_models, dead_frac=dead,
                                hidden_frac=0.05, skipcull_frac=0.02)
    if n_models == 1:
        raw["model_skip"][:] = 0
    scene = tiler.tiled_scene(raw)[0]
    cam = synth.camera(pos=(10, 5, 120))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    batch = entities.EntityBatch(scene, cuda_device)
    n = scene["n"]
    for frame in range(3):
        vis, mask = oracle_frame(scene, st, fr_o)
        batch.mq_update(fr)
        batch.compact_visible()
        check_against(batch.download(), st, vis, mask, f"{n_models} models, frame {frame}")
        dirty = (rng.uniform(0, 1, n) < 0.1) & (scene["orig_of"] >= 0)      # later frames: a tenth of the entities move
        ps = scene["pos_scale"].copy()
        ps[dirty, :3] += rng.uniform(-1, 1, (int(dirty.sum()), 3)).astype(np.float32)
        apply_frame(scene, st, (ps, scene["rot"], dirty))
        idx = np.flatnonzero(dirty)
        batch.set_transforms(idx, ps[idx], scene["rot"][idx])
    # and once with everything dirty after the partial frames (what bench.py times)
    st["flags"] |= np.where(st["flags"] & synth.E_ALIVE, synth.E_DIRTY, 0).astype(np.uint32)
    vis, mask = oracle_frame(scene, st, fr_o)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    check_against(batch.download(), st, vis, mask, f"{n_models} models, all dirty", check_flags=False)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_partial_dirty_frames(layout, cuda_device):
    """Random subsets move each frame: dirty roots drag their subtrees, clean subtrees are skipped
    (in tile layout: children of a clean parent re-read its stored matrix)."""
    from clap_amd import entities
    rng = np.random.Generator(np.random.PCG64(5))
    scene = lay_out(synth.entities_forest(6_000, 17, max_depth=7), layout)
    cam = synth.camera(pos=(0, 0, 80))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    batch = entities.EntityBatch(scene, cuda_device)
    n = scene["n"]
    for frame in range(4):
        vis, mask = oracle_frame(scene, st, fr_o)
        batch.mq_update(fr)
        batch.compact_visible()
        check_against(batch.download(), st, vis, mask, f"frame {frame}")
        dirty = (rng.uniform(0, 1, n) < 0.1) & (scene["orig_of"] >= 0)
        ps = scene["pos_scale"].copy()
        ps[dirty, :3] += rng.uniform(-5, 5, (int(dirty.sum()), 3)).astype(np.float32)
        apply_frame(scene, st, (ps, scene["rot"], dirty))
        idx = np.flatnonzero(dirty)
        batch.set_transforms(idx, ps[idx], scene["rot"][idx])


@pytest.mark.parametrize("layout", LAYOUTS)
def test_seq_wraps_like_uint16(layout, cuda_device):
    from clap_amd import entities
    scene = lay_out(synth.entities_chains(70, 3, 4), layout)
    scene["seqs"][:] = 0xFFFF | (0xFFFF << 16)
    st = ob.entity_state(scene)
    fr, _v, _p = entities.view_calc_frustum(synth.camera())
    fr_o, _vo, _po = ob.frustum_from_camera(synth.camera())
    vis, mask = oracle_frame(scene, st, fr_o)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr)
    batch.compact_visible()
    check_against(batch.download(), st, vis, mask, "wrap")
    alive = (st["flags"] & synth.E_ALIVE) != 0
    assert np.all((st["seqs"][alive] & 0xFFFF) == 0)


def test_argument_validation(cuda_device):
    import ctypes as C
    from clap_amd import _lib, entities
    scene = synth.pad_levels(synth.entities_flat(100, 1))
    batch = entities.EntityBatch(scene, cuda_device)
    L = _lib.lib()
    bad = np.asarray([0, 50, 100], np.uint32)            # level start not a multiple of 64
    rc = L.clapgpu_entities_update(None, C.byref(batch._desc), bad.ctypes.data_as(C.POINTER(C.c_uint32)), 2, 0, None)
    assert rc == _lib.ERR_INVALID_ARGUMENTS
    bad = np.asarray([0, 64], np.uint32)                 # does not end at n
    rc = L.clapgpu_entities_update(None, C.byref(batch._desc), bad.ctypes.data_as(C.POINTER(C.c_uint32)), 1, 0, None)
    assert rc == _lib.ERR_OUT_OF_BOUNDS
    rc = L.clapgpu_entities_update(None, None, None, 0, 0, None)
    assert rc == _lib.ERR_INVALID_ARGUMENTS


@pytest.mark.parametrize("layout", LAYOUTS)
def test_c2_full_size_bit_exact_and_properties(layout, cuda_device):
    """BASELINE config 2 at full size: 1M entities, depth-8 hierarchy.  The C oracle does 1M
    entities in well under a second, so the full-size check is bit-exact too; on top:
    sortedness, count == popcount(mask), and idempotence of a clean frame."""
    from clap_amd import entities
    scene = lay_out(synth.entities_chains(125_000, 8, 2), layout)
    cam = synth.camera()
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    vis, mask = oracle_frame(scene, st, fr_o)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    out = batch.download()
    check_against(out, st, vis, mask, "C2", check_flags=False)
    v = out["visible"]
    assert np.all(np.diff(v.astype(np.int64)) > 0)
    assert out["visible_count"] == int(sum(bin(int(w)).count("1") for w in out["vis_mask"]))
    assert 0.15 < out["visible_count"] / scene["n_real"] < 0.45
    mx_before = out["mx"].copy()
    batch.mq_update(fr)                                   # nothing dirty -> no rebuild
    batch.compact_visible()
    out2 = batch.download()
    assert np.array_equal(out2["mx"].view(np.uint32), mx_before.view(np.uint32))
    assert np.array_equal(out2["visible"], v)


def test_tile_external_parent(cuda_device):
    """A tile whose first row hangs off parents OUTSIDE the tile (updated by an earlier call):
    they are read from mx[] / seqs[] in HBM, the rest of the tile still chains through registers."""
    from clap_amd import entities
    base = synth.entities_chains(100, 5, 12)
    scene, tl = tiler.tiled_scene(base)
    n = scene["n"]
    cam = synth.camera()
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    vis, mask = oracle_frame(scene, st, fr_o)
    # split every tile in two: rows [0,2) form tile A, rows [2,5) form tile B whose row 0 has external parents
    trs = scene["tile_row_start"].astype(np.int64)
    top = np.stack([trs[:-1], trs[:-1] + 2], 1)
    bottom = np.stack([trs[:-1] + 2, trs[1:]], 1)
    batch = entities.EntityBatch(scene, cuda_device)
    import torch
    for part in (top, bottom):               # externals of `bottom` come from the earlier `top` calls
        # the half-tiles of one part are not contiguous in rows: launch each through a 2-entry table
        for a, b in part:
            t = torch.from_numpy(np.asarray([a, b], np.uint32).view(np.int32)).to(cuda_device)
            batch.tile_row_start, batch.n_tiles = t, 1
            batch.mq_update(fr)
    batch.compact_visible()
    check_against(batch.download(), st, vis, mask, "external parents")


def test_cull_kernel_exact_on_adversarial_boxes(cuda_device):
    """The kernels decide each frustum plane on one corner (the one maximising the fp32 dot
    chain) instead of eight.  That must give the reference's decision bit for bit, including
    boxes lying exactly on planes, planes with zero components, inverted, infinite and NaN boxes."""
    from clap_amd import entities
    rng = np.random.Generator(np.random.PCG64(123))
    n = 64 * 300
    for camkw in (dict(), dict(ndc_z_zero_one=1), dict(pos=(3, -2, 7), quat=synth.quat_from_euler_xyz(0.3, -1.1, 0.4)),
                  dict(quat=synth.quat_from_euler_xyz(0.0, np.pi / 2, 0.0))):
        cam = synth.camera(**camkw)
        fr, _v, _p = entities.view_calc_frustum(cam)
        fr_o, _vo, _po = ob.frustum_from_camera(cam)
        planes, corners = fr_o.arrays()
        lo = rng.uniform(-600, 600, (n, 3)).astype(np.float32)
        ext = rng.uniform(0, 50, (n, 3)).astype(np.float32)
        aabb = np.concatenate([lo, lo + ext], 1).astype(np.float32)
        # boxes whose corner sits (almost) exactly on a frustum corner / plane
        k = np.arange(0, n, 7)
        aabb[k, 3:6] = corners[rng.integers(0, 8, k.size), :3]
        aabb[k, 0:3] = aabb[k, 3:6] - rng.uniform(0, 1, (k.size, 3)).astype(np.float32)
        k = np.arange(3, n, 11)
        aabb[k, 0:3] = corners[rng.integers(0, 8, k.size), :3]
        aabb[k, 3:6] = np.nextafter(aabb[k, 0:3], np.float32(np.inf))
        aabb[5::97, 3] = np.inf                      # infinite
        aabb[9::101, 1] = -np.inf
        aabb[13::103, 2] = np.nan                    # NaN
        sw = np.arange(17, n, 89)                    # inverted (max < min)
        aabb[sw] = aabb[sw][:, [3, 4, 5, 0, 1, 2]]
        aabb[21::64] = 0.0                           # degenerate at the origin (camera position)
        flags = np.full(n, synth.E_ALIVE | synth.E_VISIBLE, np.uint32)
        vis, mask = ob.entities_cull(n, flags, aabb, fr_o)
        scene = synth.pad_levels(synth.entities_flat(n, 1))
        batch = entities.EntityBatch(scene, cuda_device)
        import torch
        batch.aabb.copy_(torch.from_numpy(aabb))
        batch.cull(fr)
        batch.compact_visible()
        out = batch.download()
        assert np.array_equal(out["vis_mask"], mask), camkw
        assert np.array_equal(out["visible"], vis), camkw
        assert 0 < vis.size < n
        # the same boxes through the FURTHER views' test (no early exits, per-axis extremes computed once: entities_row.h) -- this
        # camera registered four times over, next to three others: every plane of every view from one read of the boxes
        others = [synth.camera(pos=(40, 10, -30), quat=synth.quat_from_euler_xyz(-0.4, 2.2, 0.1), fov_deg=40.0, far=300.0),
                  synth.camera(pos=(-200, 5, 90), quat=synth.quat_from_euler_xyz(0.0, -np.pi / 2, 0.0), ndc_z_zero_one=1),
                  synth.camera(pos=(0, 300, 0), quat=synth.quat_from_euler_xyz(-np.pi / 2, 0.0, 0.0), aspect=1.0, near=1.0, far=700.0)]
        views = [cam] + others
        batch.set_views([entities.view_calc_frustum(c)[0] for c in views])
        batch.cull(fr)
        for v, c in enumerate(views):
            _vis_v, mask_v = ob.entities_cull(n, flags, aabb, ob.frustum_from_camera(c)[0])
            got = batch.view_masks[v].cpu().numpy().view(np.uint64)[:mask_v.size]
            assert np.array_equal(got, mask_v), (camkw, v)
        assert np.array_equal(batch.view_masks[0].cpu().numpy().view(np.uint64)[:mask.size], mask), "the main view's own planes as a further view"
        batch.set_views([])


@pytest.mark.parametrize("layout", ["levels", "tiles"])
def test_special_values_in_the_transforms(layout, cuda_device):
    """NaN / infinite / zero / denormal / huge positions, quaternions and scales, also on parents whose
    subtrees inherit them: every finite result bit-exact, NaNs where the CPU path has NaNs, and the same
    visible set (all comparisons with NaN are false on both sides)."""
    from clap_amd import entities, tiler
    base = synth.entities_chains(600, 4, seed=31)
    n = base["n"]
    rng = np.random.Generator(np.random.PCG64(31))
    ps, rot = base["pos_scale"].copy(), base["rot"].copy()
    specials = np.asarray([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 3e38, -3e38, 1e-20, 1e20], np.float32)
    victims = rng.choice(n, 400, replace=False)
    for k, e in enumerate(victims):
        v = specials[k % len(specials)]
        which = k % 3
        if which == 0:
            ps[e, rng.integers(0, 3)] = v                   # position component
        elif which == 1:
            ps[e, 3] = v                                    # scale (0 -> singular matrix -> inverse divides by 0)
        else:
            rot[e, rng.integers(0, 4)] = v                  # quaternion component
    base["pos_scale"], base["rot"] = ps, rot
    scene = synth.pad_levels(base) if layout == "levels" else tiler.tiled_scene(base)[0]
    cam = synth.camera(pos=(0, 0, 30))
    fr, _v, _p = entities.view_calc_frustum(cam)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    out = batch.download()
    st = ob.entity_state(scene)
    st["flags"] |= np.uint32(synth.E_DIRTY)
    ob.entities_update(scene, st)
    ofr, _ov, _op = ob.frustum_from_camera(cam)
    vis, _mask = ob.entities_cull(scene["n"], st["flags"], st["aabb"], ofr)
    real = scene["flags"] != 0
    for key in ("mx", "inv_mx", "aabb", "center"):
        a, b = out[key][real], st[key][real]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{key}: NaNs in the same places"
        fin = ~np.isnan(b)
        assert np.array_equal(a[fin].view(np.uint32), b[fin].view(np.uint32)), f"{key}: finite values and infinities bit-exact"
    assert np.isnan(st["mx"][real]).any() and np.isinf(st["inv_mx"][real]).any() or np.isnan(st["inv_mx"][real]).any()
    assert np.array_equal(out["visible"], vis), "visible set"
    assert 0 < len(vis) < real.sum()


def test_out_of_range_indices_from_the_host_are_harmless(cuda_device):
    """A parent / model index outside the arrays (the reference holds pointers there) must not become a
    wild device access: such an entity is updated as a root / with model 0, everything else is untouched."""
    from clap_amd import entities
    base = synth.entities_chains(200, 3, seed=3)
    scene = synth.pad_levels(base)
    n = scene["n"]
    bad_parent = np.flatnonzero(scene["parent"] >= 0)[:5]
    bad_model = np.flatnonzero(scene["flags"] != 0)[10:15]
    broken = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
    broken["parent"][bad_parent] = [n, n + 7, 2**31 - 1, 10**9, n * 3]
    broken["model"][bad_model] = [1, 2, 99, 2**30, -5]
    fixed = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
    fixed["parent"][bad_parent] = -1                         # what the guards turn them into
    fixed["model"][bad_model] = 0
    cam = synth.camera()
    fr, _v, _p = entities.view_calc_frustum(cam)
    batch = entities.EntityBatch(broken, cuda_device)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    batch.select_lod(cam["cam_pos"])
    out = batch.download()
    st = ob.entity_state(fixed)
    st["flags"] |= np.uint32(synth.E_DIRTY)
    ob.entities_update(fixed, st)
    assert_bits_equal(out["mx"], st["mx"], "mx")
    assert_bits_equal(out["aabb"], st["aabb"], "aabb")


# ---- the frame's other views culled by the same launch (clapgpu_entities.views) ------------------------------------------
# pipeline_render() runs the shadow passes with the light's view and no camera, then the model pass with the camera's
# (pipeline-builder.c:34-46, 246-272; model.c:752-760, 966-973): every view's mask and list must be what the oracle's cull
# gives for that frustum alone.
EXTRA_CAMS = [dict(pos=(40, 120, 30), quat=synth.quat_from_euler_xyz(-1.2, 0.3, 0.0), fov_deg=40.0, aspect=1.0, near=1.0, far=400.0),
              dict(pos=(-200, 10, -50), quat=synth.quat_from_euler_xyz(0.1, 1.9, 0.0)),
              dict(pos=(0, 0, 0), quat=synth.quat_from_euler_xyz(0.0, 3.1, 0.0), ndc_z_zero_one=1),
              dict(pos=(5, 400, 5), quat=synth.quat_from_euler_xyz(-1.57, 0.0, 0.0), far=90.0)]


@pytest.mark.parametrize("n_extra", [1, 4], ids=["two_frusta", "five_frusta"])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_further_views_in_the_same_launch(layout, n_extra, cuda_device):
    from clap_amd import entities
    scene = lay_out(synth.entities_forest(12_000, 5, max_depth=7, n_models=4, hidden_frac=0.1), layout)
    skip = np.flatnonzero(scene["flags"] & synth.E_ALIVE)[::17]
    scene["flags"][skip] |= synth.E_SKIP_CULLING                  # drawn by every view, whatever its planes
    cam = synth.camera(pos=(3, 4, 60))
    cams = [synth.camera(**kw) for kw in EXTRA_CAMS[:n_extra]]
    fr, _v, _p = entities.view_calc_frustum(cam)
    frs = [entities.view_calc_frustum(c)[0] for c in cams]
    fr_o = ob.frustum_from_camera(cam)[0]
    frs_o = [ob.frustum_from_camera(c)[0] for c in cams]
    st = ob.entity_state(scene)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.set_views(frs)
    rng = np.random.Generator(np.random.PCG64(11))
    for frame in range(3):
        if frame:                                               # a partial frame: most rows keep their stored boxes
            idx = rng.choice(np.flatnonzero(scene["flags"] & synth.E_ALIVE), 300, replace=False)
            scene["pos_scale"][idx, :3] += rng.uniform(-30, 30, (idx.size, 3)).astype(np.float32)
            st["flags"][idx] |= synth.E_DIRTY
            batch.set_transforms(idx, scene["pos_scale"][idx], scene["rot"][idx])
        vis, mask = oracle_frame(scene, st, fr_o)
        batch.mq_update(fr)
        batch.compact_visible()
        check_against(batch.download(), st, vis, mask, f"frame {frame} main view")
        for v, fo in enumerate(frs_o):
            vis_v, mask_v = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fo)
            got = batch.view_masks[v].cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:mask_v.size], mask_v), f"frame {frame} view {v} mask"
            batch.compact_view(v)
            out = batch.download()
            assert np.array_equal(out["visible"], vis_v), f"frame {frame} view {v} list"
            assert vis_v.size != vis.size or not np.array_equal(vis_v, vis), "a view that sees what the main one sees tests nothing"
    # the stand-alone cull pass: every plane from one read of the boxes
    for m in batch.view_masks: m.zero_()
    batch.vis_mask.zero_()
    batch.cull(fr)
    for v, fo in enumerate(frs_o):
        _vis_v, mask_v = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fo)
        assert np.array_equal(batch.view_masks[v].cpu().numpy().view(np.uint64)[:mask_v.size], mask_v), f"cull pass, view {v}"
    assert np.array_equal(batch.download()["vis_mask"][:mask.size], mask)
    # off again: the planes are left alone
    batch.set_views([])
    batch.mq_update(fr)
    batch.compact_visible()
    assert np.array_equal(batch.download()["visible"], vis)


def test_views_argument_validation(cuda_device):
    from clap_amd import entities, _lib
    scene = synth.pad_levels(synth.entities_flat(200, 3))
    batch = entities.EntityBatch(scene, cuda_device)
    fr = entities.view_calc_frustum(synth.camera())[0]
    batch.set_views([fr])
    batch._views.n = _lib.EXTRA_VIEWS_MAX + 1
    with pytest.raises(_lib.ClapGpuError):
        batch.mq_update(fr)
    batch._views.n = 1
    batch._views.vis_mask[0] = None
    with pytest.raises(_lib.ClapGpuError):
        batch.mq_update(fr)
