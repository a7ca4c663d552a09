"""GPU: BASELINE configs[4] (8 x MI355X: 16 M entities + 2 M particles sharded by range) as far as one GPU can run it.

* one RANK of it at full per-rank size -- 250 000 chains x depth 8 = 2 M entities + 262 144 particles -- through the very
  object bench.py's timed loop steps (bench.RankStep), exchange on (RCCL ncclAllGather of the visibility mask from C,
  world size 1), against the oracle: transforms, boxes, the gathered visible set, the particle stream and its drand48 state;
* the 16 M-id global expansion every rank runs behind the allgather: clapgpu_visible_compact over a gathered mask of
  8 ranks x n_pad entities (the two-launch count + expand path, > 4 M entities) against numpy, and rank by rank with
  index_base = r * n_pad.
The N-rank launch itself is tests/test_bench_launcher.py (CPU); the N-rank exchange logic tests/test_shard_cpu.py (gloo)."""
import ctypes as C
import numpy as np
import pytest

from oracle import binding as ob
from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu

C5_CHAINS, C5_DEPTH, C5_PARTICLES, C5_WORLD = 250_000, 8, 262_144, 8


def test_one_rank_of_c5_with_exchange_matches_oracle(cuda_device, process_group):
    import torch
    import bench
    # block 5 of the 8 (its seeds, its particle stream) in a world of one rank
    rs = bench.RankStep(C5_CHAINS, C5_DEPTH, C5_PARTICLES, 0, 1, cuda_device, use_dist=True, route="rccl", block=5)
    assert rs.index_base == 5 * rs.batch.n
    try:
        assert rs.xch.direct is not None, "the direct RCCL communicator must come up on the GPU box"
        assert rs.batch.n_real == 2_000_000 and rs.pbatch.n_real == C5_PARTICLES
        frames = 3
        for _ in range(frames):
            rs.step()
        torch.cuda.synchronize()
        cnt, ids = rs.xch.last()
        got_vis = ids[:int(cnt.item())].cpu().numpy().view(np.uint32)
        out = rs.batch.download()
        pout = rs.pbatch.download()
    finally:
        rs.xch.destroy()

    # ---- entities: the oracle on the same 2 M-entity block ----
    scene = rs.scene
    fr, _v, _p = ob.frustum_from_camera(rs.cam)
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    vis, mask = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr)
    for k in ("mx", "inv_mx", "aabb", "center"):
        assert_bits_equal(out[k], st[k], f"c5 rank share: {k}")
    assert vis.size > 100_000
    assert np.array_equal(got_vis, vis), "gathered visible set (world 1) differs from the oracle's list"
    assert np.all(np.diff(got_vis.astype(np.int64)) > 0)

    # ---- particles: the rank's own drand48 stream, bit for bit ----
    pos, vel, pst = ob.particles_spawn(rs.psys, rs.pstate0)
    for _ in range(frames):
        _k, pst = ob.particles_update(rs.psys, pos, vel, pst)
    assert_bits_equal(pout["pos"], pos, "c5 rank share: particle pos_array")
    assert_bits_equal(pout["vel"], vel, "c5 rank share: particle velocity")
    assert pout["rng_state"] == pst


def _compact(L, mask_t, n, base, out_t, cnt_t, scratch_t, row_pop=None):
    from clap_amd import _lib
    rc = L.clapgpu_visible_compact(None, mask_t.data_ptr(), None if row_pop is None else row_pop.data_ptr(), n, base,
                                   out_t.data_ptr(), cnt_t.data_ptr(), scratch_t.data_ptr())
    _lib.check(rc, "clapgpu_visible_compact")


def test_16m_id_global_expansion_matches_numpy(cuda_device):
    """The mask 8 ranks x 2 M entities gather into (2 MB) expands into the ascending list of up to 16 M global ids."""
    import torch
    from clap_amd import _lib
    L = _lib.lib()
    # n_pad of a real c5 block without building one: 64 chains per tile, one 64-entity row per level (C2: 1 000 448)
    n_pad = ((C5_CHAINS + 63) // 64) * 64 * C5_DEPTH
    assert n_pad == 2_000_384
    words = n_pad // 64
    n_all = C5_WORLD * n_pad
    assert n_all > (1 << 22), "must take the two-launch path"
    rng = np.random.Generator(np.random.PCG64(55))
    g = np.zeros(C5_WORLD * words, np.uint64)
    for r in range(C5_WORLD):
        blk = g[r * words:(r + 1) * words]
        if r == 2:
            continue                                         # a rank that sees nothing
        if r == 6:
            blk[:] = ~np.uint64(0)                           # a rank that sees everything, padding slots included
            continue
        dens = (0.3, 0.02, 0.0, 0.9, 0.5, 0.3, 1.0, 0.3)[r]
        bits = rng.random(n_pad) < dens
        blk[:] = np.packbits(bits, bitorder="little").view(np.uint64)
    g[words - 1] = np.uint64(1) << np.uint64(63)             # last slot of rank 0, first of rank 1
    g[words] |= np.uint64(1)
    expect = np.flatnonzero(np.unpackbits(g.view(np.uint8), bitorder="little")).astype(np.uint32)

    mask_t = torch.from_numpy(g.view(np.int64)).to(cuda_device)
    out_t = torch.zeros(n_all, dtype=torch.int32, device=cuda_device)
    cnt_t = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    scratch_t = torch.zeros(L.clapgpu_visible_scratch_bytes(n_all) // 4 + 4, dtype=torch.int32, device=cuda_device)
    _compact(L, mask_t, n_all, 0, out_t, cnt_t, scratch_t)
    torch.cuda.synchronize()
    total = int(cnt_t.item())
    assert total == expect.size and total > 6_000_000
    got = out_t[:total].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, expect), "global id list differs from numpy"

    # rank by rank: the slice a rank contributes, expanded with its index base, is its slice of the global list
    per_t = torch.zeros(n_pad, dtype=torch.int32, device=cuda_device)
    at = 0
    for r in range(C5_WORLD):
        sl = mask_t[r * words:(r + 1) * words]
        _compact(L, sl, n_pad, r * n_pad, per_t, cnt_t, scratch_t)
        torch.cuda.synchronize()
        c = int(cnt_t.item())
        assert np.array_equal(per_t[:c].cpu().numpy().view(np.uint32), expect[at:at + c]), f"rank {r} slice"
        at += c
    assert at == total

    # n not a multiple of the 4096-entity group, nor of 64: the tail bits past n are padding
    n_odd = n_all - 4096 - 37
    _compact(L, mask_t, n_odd, 0, out_t, cnt_t, scratch_t)
    torch.cuda.synchronize()
    e2 = expect[expect < n_odd]
    assert int(cnt_t.item()) == e2.size
    assert np.array_equal(out_t[:e2.size].cpu().numpy().view(np.uint32), e2)


def test_bench_line_carries_the_configs4_leg_when_distributed(tmp_path):
    """`bench.py --gpus N` with N > 1 (here: a world of one with the collective path forced, CLAP_BENCH_FORCE_DIST=1 -- the
    same code the ranks of an 8-GPU run execute) times TWO legs in one process group over one communicator: the weak-scaling
    headline (configs[1] per GPU) and BASELINE configs[4]'s per-rank share, reported under `c5`; `summary` and `c5` precede
    the long sections of the line."""
    import json
    import os
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(CLAP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-frames", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["value"] > 0
    assert r["config"]["entities_per_gpu"] == 1_000_000, "the headline stays configs[1] per GPU"
    c5 = r["c5"]
    assert list(c5) == bench.C5_KEYS
    assert c5["entities_per_gpu"] == 2_000_000 and c5["particles_per_gpu"] == C5_PARTICLES and c5["value"] > 0 and c5["visible"] > 100_000
    assert c5["global_ids"] >= 2_000_000 and c5["ms_per_step"] > 0 and c5["particle_updates_per_s"] > 0
    keys = list(r)
    assert keys.index("summary") < keys.index("c5") < keys.index("config") < keys.index("roofline")
