"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / clap_amd.shard -- range sharding by
whole tiles and the allgather of the compacted visible set -- using the oracle as the per-rank
"device".  (The HIP kernels themselves are covered single-GPU; the exchange is what N > 1 adds.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clap_amd import shard, synth, tiler


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        # one global scene, tiled; each rank updates its contiguous tile range only
        scene, tl = tiler.tiled_scene(synth.entities_forest(6000, seed=31))
        trs = scene["tile_row_start"].astype(np.int64)
        t0, t1 = shard.shard_tile_ranges(np.diff(trs), world)[rank]
        lo, hi = int(trs[t0]) * 64, int(trs[t1]) * 64
        cam = synth.camera(pos=(0, 5, 60))
        fr, _v, _p = ob.frustum_from_camera(cam)
        sub = {k: (scene[k][lo:hi].copy() if isinstance(scene[k], np.ndarray) and scene[k].shape[:1] == (scene["n"],)
                   else scene[k]) for k in scene}
        sub["n"] = hi - lo
        sub["parent"] = np.where(sub["parent"] >= 0, sub["parent"] - lo, -1).astype(np.int32)
        assert sub["parent"].max(initial=-1) < sub["n"] and np.all(sub["parent"][sub["parent"] >= 0] >= 0), \
            "a subtree crosses a shard border"
        st = ob.entity_state(sub)
        ob.entities_update(sub, st)
        vis, _mask = ob.entities_cull(sub["n"], st["flags"], st["aabb"], fr)
        vis_global = torch.zeros(max(sub["n"], 1), dtype=torch.int32)
        vis_global[:len(vis)] = torch.from_numpy((vis.astype(np.int64) + lo).astype(np.int32))
        counts, gathered = shard.allgather_visible(vis_global, torch.tensor([len(vis)], dtype=torch.int32), world,
                                                   pad_to=64)
        full = shard.concat_visible(counts, gathered).numpy()
        # single-process answer
        st_all = ob.entity_state(scene)
        ob.entities_update(scene, st_all)
        vis_all, _m = ob.entities_cull(scene["n"], st_all["flags"], st_all["aabb"], fr)
        ok = np.array_equal(full.astype(np.uint32), vis_all) and bool(np.all(np.diff(full) > 0))
        # the form bench.py uses: one fixed-size allgather of the 1-bit-per-entity mask (equal shard
        # capacity on every rank), expanded locally into the same global list
        cap_words = (scene["n"] + 63) // 64                       # common capacity: the whole scene
        local = np.zeros(cap_words, np.uint64)
        local[:len(_mask)] = _mask
        g = shard.allgather_visible_mask(torch.from_numpy(local.view(np.int64)), world).numpy().view(np.uint64)
        los = [int(trs[a]) * 64 for a, _b in shard.shard_tile_ranges(np.diff(trs), world)]
        ids = []
        for r in range(world):
            bits = np.unpackbits(g[r * cap_words:(r + 1) * cap_words].view(np.uint8), bitorder="little")
            ids.append(np.flatnonzero(bits) + los[r])
        ok = ok and np.array_equal(np.concatenate(ids).astype(np.uint32), vis_all)
        q.put((rank, ok, int(counts.sum()), len(vis_all), (t0, t1)))
    finally:
        dist.destroy_process_group()


def test_sharded_update_and_visible_allgather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, total, expected, rng in sorted(res):
        assert ok, f"rank {rank}: gathered visible set differs from the single-process result"
        assert total == expected and expected > 0
    assert sorted(r[4] for r in res)[0][0] == 0


def _worker_ranges(rank, world, port, q):
    """World 8, UNEVEN shards, one of them EMPTY: the masks travel at the common capacity and are expanded with per-rank
    bases -- clapgpu_shard_bases + clapgpu_visible_expand_ranges_host, the id arithmetic of clapgpu_exchange_visible_ranges."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        scene, tl = tiler.tiled_scene(synth.entities_forest(1100, seed=77, max_depth=7))
        trs = scene["tile_row_start"].astype(np.int64)
        n_tiles = len(trs) - 1
        cuts = shard.shard_tile_ranges(np.diff(trs), world)
        base, n_pad, cap_pad = shard.shard_bases(trs, world)
        t0, t1 = cuts[rank]
        lo, hi = int(trs[t0]) * 64, int(trs[t1]) * 64
        assert base[rank] == lo and n_pad[rank] == hi - lo and cap_pad == max(64, int(n_pad.max()))
        cam = synth.camera(pos=(0, 5, 60))
        fr, _v, _p = ob.frustum_from_camera(cam)
        local = np.zeros(cap_pad // 64, np.uint64)               # zero beyond this rank's own words (all of it on an empty rank)
        n_vis = 0
        if hi > lo:
            sub = {k: (scene[k][lo:hi].copy() if isinstance(scene[k], np.ndarray) and scene[k].shape[:1] == (scene["n"],)
                       else scene[k]) for k in scene}
            sub["n"] = hi - lo
            sub["parent"] = np.where(sub["parent"] >= 0, sub["parent"] - lo, -1).astype(np.int32)
            assert np.all(sub["parent"][sub["parent"] >= 0] < sub["n"]), "a subtree crosses a shard border"
            st = ob.entity_state(sub)
            ob.entities_update(sub, st)
            vis, mask = ob.entities_cull(sub["n"], st["flags"], st["aabb"], fr)
            local[:len(mask)] = mask
            n_vis = len(vis)
        g = shard.allgather_visible_mask(torch.from_numpy(local.view(np.int64)), world).numpy().view(np.uint64)
        ids = shard.expand_ranges_host(g, world, cap_pad, base, n_pad)
        st_all = ob.entity_state(scene)
        ob.entities_update(scene, st_all)
        vis_all, _m = ob.entities_cull(scene["n"], st_all["flags"], st_all["aabb"], fr)
        ok = np.array_equal(ids, vis_all.astype(np.uint32)) and bool(np.all(np.diff(ids.astype(np.int64)) > 0))
        q.put((rank, ok, n_vis, len(vis_all), int(n_pad[rank]), n_tiles, int(cap_pad)))
    finally:
        dist.destroy_process_group()


def test_uneven_and_empty_shards_allgather_world8():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ranges, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [r[4] for r in res]
    assert all(r[1] for r in res), "a rank's expanded global list differs from the single-process visible set"
    assert sum(r[2] for r in res) == res[0][3] > 0
    assert min(sizes) == 0, f"one shard was to be empty: {sizes} ({res[0][5]} tiles)"
    assert len(set(s for s in sizes if s)) > 1, f"shards were to be uneven: {sizes}"
    assert res[0][6] == max(sizes)


def test_shard_tile_ranges_cover_and_balance():
    rows = np.asarray([8] * 100 + [1] * 37 + [3] * 11)
    for world in (1, 2, 3, 4, 8):
        rngs = shard.shard_tile_ranges(rows, world)
        assert rngs[0][0] == 0 and rngs[-1][1] == len(rows)
        assert all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
        loads = [int(rows[a:b].sum()) for a, b in rngs]
        assert max(loads) - min(loads) <= 16


def test_c_shard_ranges_cover_and_balance():
    """clapgpu_shard_tile_range (exchange.hip, host code): contiguous, complete, balanced to within one tile, identical
    on every rank; edge cases: more ranks than tiles, a single tile, equal tiles."""
    rng = np.random.Generator(np.random.PCG64(3))
    for rows in ([8] * 16, [1], rng.integers(1, 9, 1000).tolist(), [5, 1, 1, 1, 9, 2], [3] * 5):
        for world in (1, 2, 3, 4, 8):
            r = shard.shard_tile_ranges(rows, world)
            assert r[0][0] == 0 and r[-1][1] == len(rows)
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1)), "contiguous"
            assert all(a <= b for a, b in r)
            if len(rows) >= 4 * world:
                per = [sum(rows[a:b]) for a, b in r]
                assert max(per) - min(per) <= 2 * max(rows), (rows[:8], world, per)


def test_blocks_of_the_global_forest_are_the_shards():
    """bench.py --gpus N builds rank r's scene as block r of ONE global forest (synth.entities_chains, seed 2 + r).
    Tiling the concatenated forest and cutting it with clapgpu_shard_tile_range gives every rank exactly its block: same
    rows, same entities, and the union of the shard-local visible sets (ids + rank * n_pad) is the global visible set."""
    from oracle import binding as ob
    world, chains, depth = 4, 256, 4
    blocks = [tiler.tiled_scene(synth.entities_chains(chains, depth, seed=2 + r))[0] for r in range(world)]
    n_pad = blocks[0]["n"]
    assert all(b["n"] == n_pad for b in blocks), "equal padded shard sizes"
    trs = np.concatenate([[0]] + [b["tile_row_start"][1:].astype(np.int64) + r * (n_pad // 64) for r, b in enumerate(blocks)])
    cuts = shard.shard_tile_ranges(np.diff(trs), world)
    tiles_per_block = len(blocks[0]["tile_row_start"]) - 1
    assert cuts == [(r * tiles_per_block, (r + 1) * tiles_per_block) for r in range(world)]
    cam = synth.camera(pos=(0, 5, 60))
    fr, _v, _p = ob.frustum_from_camera(cam)
    union = []
    for r, b in enumerate(blocks):
        st = ob.entity_state(b)
        ob.entities_update(b, st)
        vis, _m = ob.entities_cull(b["n"], st["flags"], st["aabb"], fr)
        union.append(vis.astype(np.int64) + r * n_pad)
    union = np.concatenate(union)
    # the global forest as one scene: parents shifted by the block offset
    g = {k: (np.concatenate([b[k] for b in blocks]) if isinstance(blocks[0][k], np.ndarray) and blocks[0][k].shape[:1] == (n_pad,)
             else blocks[0][k]) for k in blocks[0]}
    g["parent"] = np.concatenate([np.where(b["parent"] >= 0, b["parent"] + r * n_pad, -1) for r, b in enumerate(blocks)]).astype(np.int32)
    g["n"] = world * n_pad
    st = ob.entity_state(g)
    ob.entities_update(g, st)
    vis_all, _m = ob.entities_cull(g["n"], st["flags"], st["aabb"], fr)
    assert np.array_equal(union, vis_all.astype(np.int64)) and np.all(np.diff(union) > 0)
