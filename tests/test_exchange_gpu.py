"""GPU, world size 1: the path's one exchange exactly as bench.py --gpus N runs it (clap_amd.shard.VisibleExchange):
libclapgpu's clapgpu_exchange (RCCL opened at run time from C, unique id carried by the process group),
clapgpu_exchange_visible = ncclAllGather of the visibility mask + expansion on the side stream, and
clapgpu_visible_compact over a gathered mask with a non-zero index base -- against the oracle's visible list.
(N > 1 on hardware is the driver's scaling run; the multi-rank logic is covered on CPU by test_shard_cpu.py.)"""
import numpy as np
import pytest

from clap_amd import synth, tiler
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _oracle_visible(scene, cam):
    fr, _v, _p = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    vis, mask = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr)
    return vis, mask


@pytest.mark.parametrize("route", ["rccl", "c10d"])
def test_world1_exchange_matches_oracle(route, cuda_device, process_group):
    import torch
    from clap_amd import entities, shard
    scene, _tl = tiler.tiled_scene(synth.entities_forest(20_000, seed=77))
    cams = [synth.camera(pos=(0, 5, 60)), synth.camera(pos=(30, 0, -20)), synth.camera(pos=(0, 0, 0))]
    batch = entities.EntityBatch(scene, cuda_device)
    xch = shard.VisibleExchange(batch, 0, 1, cuda_device, route=route)
    if route == "rccl":
        assert xch.direct is not None, "direct RCCL communicator must come up on the GPU box"
        assert "clapgpu_exchange_visible" in xch.route
    else:
        assert xch.direct is None
    # the line bench.py prints proves its own N: what the communicator says about itself + the device behind each rank
    import bench
    proof = xch.proof()
    assert proof["rccl_ranks"] == 1 and proof["comm_rank_ok"] and len(proof["devices"]) == 1 and proof["devices"][0]
    assert bench.check_proof(proof, 1) is None
    assert "communicator has 1 ranks" in bench.check_proof(proof, 8), "a world of one must not pass for --gpus 8"
    if route == "rccl":
        assert ":" in proof["devices"][0], f"a PCI bus id, got {proof['devices'][0]!r}"
    try:
        for f, cam in enumerate(cams * 2):                  # six frames: both mask buffers reused several times
            fr, _v, _p = entities.view_calc_frustum(cam)
            xch.begin()
            batch.mq_update(fr, all_dirty=True)
            xch.submit()
            torch.cuda.synchronize()
            cnt, ids = xch.last()
            got = ids[:int(cnt.item())].cpu().numpy().view(np.uint32)
            vis, mask = _oracle_visible(scene, cam)
            assert np.array_equal(got, vis), f"frame {f} ({route}): gathered visible set differs from the oracle"
            g = xch.g_mask[(xch.frame - 1) & 1].cpu().numpy().view(np.uint64)
            assert np.array_equal(g[:len(mask)], mask)
    finally:
        xch.destroy()


def test_gathered_mask_expands_with_index_base(cuda_device, process_group):
    """What rank r > 0 contributes: its ids offset by its range start.  Emulated on one GPU: clapgpu_exchange_visible in
    its mask-only form (visible = NULL), then clapgpu_visible_compact over the gathered mask with index_base = r * n_pad,
    the id arithmetic of exchange.hip / bench.py."""
    import ctypes as C
    import torch
    from clap_amd import _lib, entities, shard
    scene, _tl = tiler.tiled_scene(synth.entities_chains(700, 5, seed=5))
    cam = synth.camera(pos=(0, 0, 40))
    batch = entities.EntityBatch(scene, cuda_device)
    fr, _v, _p = entities.view_calc_frustum(cam)
    batch.mq_update(fr, all_dirty=True)
    xch = shard.VisibleExchange(batch, 0, 1, cuda_device, route="rccl")
    assert xch.direct is not None
    side = torch.cuda.Stream(device=cuda_device)
    gathered = torch.zeros_like(batch.vis_mask)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    L = _lib.lib()
    try:
        with torch.cuda.stream(side):
            side.wait_event(ev)
            _lib.check(L.clapgpu_exchange_visible(C.c_void_p(side.cuda_stream), xch.direct, batch.vis_mask.data_ptr(), batch.n,
                                                  gathered.data_ptr(), None, None, None), "clapgpu_exchange_visible(mask only)")
            out = torch.zeros(batch.n, dtype=torch.int32, device=cuda_device)
            cnt = torch.zeros(1, dtype=torch.int32, device=cuda_device)
            scratch = torch.zeros(L.clapgpu_visible_scratch_bytes(batch.n) // 4 + 4, dtype=torch.int32, device=cuda_device)
            base = 3 * batch.n
            rc = L.clapgpu_visible_compact(C.c_void_p(side.cuda_stream), gathered.data_ptr(), None, batch.n,
                                           base, out.data_ptr(), cnt.data_ptr(), scratch.data_ptr())
            _lib.check(rc, "clapgpu_visible_compact")
        torch.cuda.synchronize()
    finally:
        xch.destroy()
    vis, _mask = _oracle_visible(scene, cam)
    got = out[:int(cnt.item())].cpu().numpy().astype(np.int64)
    assert np.array_equal(got, vis.astype(np.int64) + base)


def test_gathered_masks_of_uneven_shards_expand_to_scene_global_ids(cuda_device):
    """The second half of clapgpu_exchange_visible_ranges on one GPU: eight ranks' masks as a gathered array -- uneven
    shards cut by clapgpu_shard_tile_range, two of them empty, every segment padded to the common capacity, garbage
    beyond each rank's own words (a sender zeroes them; the expansion must not depend on it) -- expanded by
    clapgpu_visible_compact_ranges into the ascending scene-global list, against the host twin
    (clapgpu_visible_expand_ranges_host, which test_shard_cpu.py runs over eight gloo ranks) and the oracle's
    single-scene visible set."""
    import ctypes as C
    import torch
    from clap_amd import _lib, shard
    world = 8
    scene, _tl = tiler.tiled_scene(synth.entities_forest(1100, seed=77, max_depth=7))
    trs = scene["tile_row_start"].astype(np.int64)
    base, n_pad, cap_pad = shard.shard_bases(trs, world)
    assert (n_pad == 0).any() and len(set(int(x) for x in n_pad if x)) > 1
    cam = synth.camera(pos=(0, 5, 60))
    vis, mask = _oracle_visible(scene, cam)
    cw = cap_pad // 64
    rng = np.random.default_rng(3)
    g = rng.integers(0, 2 ** 63, world * cw, dtype=np.int64).view(np.uint64)       # garbage everywhere ...
    for r in range(world):
        w0, nw = int(base[r]) // 64, int(n_pad[r]) // 64
        g[r * cw:r * cw + nw] = mask[w0:w0 + nw]                                    # ... but each rank's own words
    host = shard.expand_ranges_host(g, world, cap_pad, base, n_pad)
    assert np.array_equal(host, vis.astype(np.uint32))
    L = _lib.lib()
    d_g = torch.from_numpy(g.view(np.int64).copy()).to(cuda_device)
    out = torch.zeros(world * cap_pad, dtype=torch.int32, device=cuda_device)
    cnt = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    scratch = torch.zeros(L.clapgpu_visible_scratch_bytes(world * cap_pad) // 4 + 4, dtype=torch.int32, device=cuda_device)
    rc = L.clapgpu_visible_compact_ranges(C.c_void_p(torch.cuda.current_stream().cuda_stream), d_g.data_ptr(), world, cap_pad,
                                          base.ctypes.data, n_pad.ctypes.data, out.data_ptr(), cnt.data_ptr(), scratch.data_ptr())
    _lib.check(rc, "clapgpu_visible_compact_ranges")
    torch.cuda.synchronize()
    got = out[:int(cnt.item())].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, vis.astype(np.uint32))
    # refusals: overlapping or descending ranges, a size that is not a whole row
    bad = base.copy(); bad[2] = bad[0]
    assert L.clapgpu_visible_compact_ranges(None, d_g.data_ptr(), world, cap_pad, bad.ctypes.data, n_pad.ctypes.data, out.data_ptr(),
                                            cnt.data_ptr(), scratch.data_ptr()) == _lib.ERR_INVALID_ARGUMENTS
    odd = n_pad.copy(); odd[0] += 1
    assert L.clapgpu_visible_compact_ranges(None, d_g.data_ptr(), world, cap_pad, base.ctypes.data, odd.ctypes.data, out.data_ptr(),
                                            cnt.data_ptr(), scratch.data_ptr()) == _lib.ERR_INVALID_ARGUMENTS
