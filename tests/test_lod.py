"""Per-pass LOD selection (model.c:975-992).  CPU: the oracle's building
blocks against the reference (entity3d_aabb_avg_edge incl. libm cbrtf, entity3d_set_lod).
GPU: the kernel against the oracle -- integer LODs, bit-exact."""
import numpy as np
import pytest

from clap_amd import synth, tiler
from oracle import binding as ob
from oracle import refrun


@pytest.mark.skipif(not refrun.available(), reason="reference build (oracle/_ref) not present")
def test_oracle_blocks_match_reference():
    rng = np.random.Generator(np.random.PCG64(1))
    n = 4000
    aabb = np.concatenate([-rng.uniform(0.05, 6, (n, 3)), rng.uniform(0.05, 6, (n, 3))], 1).astype(np.float32)
    scale = rng.uniform(0.1, 4, n).astype(np.float32)
    lmin = rng.integers(0, 2, n)
    lmax = lmin + rng.integers(0, 3, n)
    req = rng.integers(-3, 10, n)
    edge, lod = refrun.lod_blocks(aabb, scale, lmin, lmax, req)
    mine = np.asarray([ob.lib().clapo_aabb_avg_edge(aabb[i], float(scale[i])) for i in range(n)], np.float32)
    assert np.array_equal(mine.view(np.uint32), edge.view(np.uint32)), "entity3d_aabb_avg_edge"
    assert np.array_equal(np.clip(req, lmin, lmax), lod), "entity3d_set_lod(e, lod, false)"


def _lod_scene(layout):
    base = synth.entities_forest(30_000, seed=41, n_models=4, max_depth=5)
    base["model_skip"][:] = 0
    scene = synth.pad_levels(base) if layout == "levels" else tiler.tiled_scene(base)[0]
    scene["model_lod"] = np.asarray([[0, 3], [1, 2], [0, 0], [0, 3]], np.uint8)
    return scene


def test_oracle_lod_semantics():
    scene = _lod_scene("levels")
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    fr, _v, _p = ob.frustum_from_camera(synth.camera(pos=(0, 0, 0)))
    vis, _m = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr)
    cur = np.full(scene["n"], 7, np.int32)
    force = np.full(scene["n"], -1, np.int32)
    force[vis[::10]] = 2
    draw = ob.entities_lod(scene, st, vis, (0, 0, 0), scene["model_lod"], force, cur)
    assert np.all(draw[::10] == 2), "a forced LOD wins"
    lo, hi = scene["model_lod"][scene["model"][vis], 0], scene["model_lod"][scene["model"][vis], 1]
    free = np.ones(len(vis), bool); free[::10] = False
    inside = np.all((st["aabb"][vis, :3] <= 0) & (st["aabb"][vis, 3:] >= 0), axis=1)
    assert np.all((draw[free & ~inside] >= lo[free & ~inside]) & (draw[free & ~inside] <= hi[free & ~inside]))
    assert np.all(draw[free & inside] == 7), "camera inside the box: cur_lod is kept"
    assert len(np.unique(draw)) >= 3
    hidden = np.setdiff1d(np.arange(scene["n"]), vis)
    assert np.all(cur[hidden] == 7), "culled entities keep their LOD"


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["levels", "tiles"])
def test_hip_lod_matches_oracle(layout, cuda_device):
    from clap_amd import entities
    scene = _lod_scene(layout)
    cam = synth.camera(pos=(30, 5, 40))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    vis, _m = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr_o)
    rng = np.random.Generator(np.random.PCG64(2))
    force = np.where(rng.uniform(0, 1, scene["n"]) < 0.1, rng.integers(0, 4, scene["n"]), -1).astype(np.int32)
    cur = np.zeros(scene["n"], np.int32)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr)
    batch.compact_visible()
    for cam_pos in ((30, 5, 40), (-100, 20, 300), tuple(st["center"][vis[5]])):      # last: camera inside a box
        draw = ob.entities_lod(scene, st, vis, cam_pos, scene["model_lod"], force, cur)
        batch.select_lod(cam_pos, force)
        import torch
        torch.cuda.synchronize()
        got_draw = batch.draw_lod[:len(vis)].cpu().numpy()
        assert np.array_equal(got_draw, draw), f"draw list LODs, camera {cam_pos}"
        assert np.array_equal(batch.cur_lod.cpu().numpy()[:scene["n"]], cur), "entity3d.cur_lod"
    assert len(np.unique(draw)) >= 3
    # the list and its LODs by one call (clapgpu_visible_compact_lod, what a frame issues): the same list, the same picks
    fused = entities.EntityBatch(scene, cuda_device)
    fused.mq_update(fr)
    cur2 = np.zeros(scene["n"], np.int32)
    for cam_pos in ((30, 5, 40), (-100, 20, 300), tuple(st["center"][vis[5]])):
        draw = ob.entities_lod(scene, st, vis, cam_pos, scene["model_lod"], force, cur2)
        fused.visible.fill_(-1)
        fused.compact_visible_lod(cam_pos, force)
        torch.cuda.synchronize()
        assert int(fused.visible_count.item()) == len(vis)
        assert np.array_equal(fused.visible[:len(vis)].cpu().numpy().view(np.uint32), vis), "the fused launch's list"
        assert np.array_equal(fused.draw_lod[:len(vis)].cpu().numpy(), draw), f"the fused launch's LODs, camera {cam_pos}"
        assert np.array_equal(fused.cur_lod.cpu().numpy()[:scene["n"]], cur2)
