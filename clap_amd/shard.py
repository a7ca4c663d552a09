"""Multi-GPU sharding of the scene update: one process per GPU, contiguous tile ranges.

The path shards by entity range with whole subtrees inside one shard, so the update itself
needs no collective.  The single exchange is the allgather of each shard's compacted visible
list (global entity ids): afterwards every rank holds the identical ascending visible set.
``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs under "gloo" on CPU
tensors (tests/test_shard_cpu.py, world_size 2).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_tile_ranges(rows_per_tile, world):
    """Split tiles into `world` contiguous ranges of (nearly) equal row count (clapgpu_shard_tile_range, the C
    function every rank of an engine would call).  rows_per_tile: rows of each tile (tile_row_start differences).
    Returns [(t0, t1)] per rank."""
    import ctypes as C
    from . import _lib
    rows = np.asarray(rows_per_tile, np.int64)
    trs = np.ascontiguousarray(np.concatenate([[0], np.cumsum(rows)]), np.uint32)
    out = []
    for r in range(world):
        t0, t1 = C.c_uint32(), C.c_uint32()
        _lib.check(_lib.lib().clapgpu_shard_tile_range(trs.ctypes.data, len(rows), r, world, C.byref(t0), C.byref(t1)),
                   "clapgpu_shard_tile_range")
        out.append((t0.value, t1.value))
    return out


def shard_bases(tile_row_start, world):
    """(base[world], n_pad[world], cap_pad) of the cut shard_tile_ranges() makes: every rank's first scene-global id, its
    padded size and the largest size (clapgpu_shard_bases, host C: the same arithmetic every rank runs)."""
    import ctypes as C
    from . import _lib
    trs = np.ascontiguousarray(tile_row_start, np.uint32)
    base, n_pad, cap = np.zeros(world, np.uint32), np.zeros(world, np.uint32), C.c_uint32()
    _lib.check(_lib.lib().clapgpu_shard_bases(trs.ctypes.data, len(trs) - 1, world, base.ctypes.data, n_pad.ctypes.data,
                                              C.byref(cap)), "clapgpu_shard_bases")
    return base, n_pad, int(cap.value)


def expand_ranges_host(gathered_mask, world, cap_pad, base, n_pad):
    """The gathered per-rank masks (world * cap_pad / 64 words, rank-major) as the ascending scene-global id list:
    clapgpu_visible_expand_ranges_host, the host twin of the expansion kernel (same id arithmetic)."""
    from . import _lib
    g = np.ascontiguousarray(gathered_mask).view(np.uint64)
    base, n_pad = np.ascontiguousarray(base, np.uint32), np.ascontiguousarray(n_pad, np.uint32)
    L = _lib.lib()
    n = L.clapgpu_visible_expand_ranges_host(g.ctypes.data, world, cap_pad, base.ctypes.data, n_pad.ctypes.data, None, 0)
    out = np.zeros(max(int(n), 1), np.uint32)
    got = L.clapgpu_visible_expand_ranges_host(g.ctypes.data, world, cap_pad, base.ctypes.data, n_pad.ctypes.data, out.ctypes.data, int(n))
    assert got == n
    return out[:n]


def allgather_visible(visible, count, world, counts_buf=None, gather_buf=None, pad_to=4096, group=None):
    """visible: this rank's ascending global ids (capacity >= its count), count: 1-element tensor.
    Returns (counts[world] on device, gathered[world, cap] tensor); rank r's ids are
    gathered[r, :counts[r]].  One small and one payload allgather; the payload is padded to the
    largest count (RCCL has no allgatherv), rounded up to `pad_to`."""
    if counts_buf is None:
        counts_buf = torch.empty(world, dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(counts_buf, count, group=group)
    cap = int(counts_buf.max().item())
    cap = max((cap + pad_to - 1) // pad_to * pad_to, pad_to)
    cap = min(cap, visible.shape[0])
    if gather_buf is None or gather_buf.numel() < world * cap:
        gather_buf = torch.empty(world * cap, dtype=visible.dtype, device=visible.device)
    out = gather_buf[:world * cap]
    dist.all_gather_into_tensor(out, visible[:cap].contiguous(), group=group)
    return counts_buf, out.view(world, cap)


def allgather_visible_mask(vis_mask, world, out=None, group=None):
    """The compacted visible set as its 1-bit-per-entity mask: ONE fixed-size allgather, no counts,
    no padding, no host sync, ~9x fewer bytes than the id list at 30 % visibility.  Rank r's
    words land at out[r * n_words : (r + 1) * n_words], i.e. `out` is the visibility mask of the
    global entity range when every shard has the same (padded) size; expanding it locally
    (clapgpu_visible_compact over world * n entities) gives every rank the identical ascending
    global id list."""
    n_words = vis_mask.numel()
    if out is None:
        out = torch.empty(world * n_words, dtype=vis_mask.dtype, device=vis_mask.device)
    dist.all_gather_into_tensor(out[:world * n_words], vis_mask, group=group)
    return out


def concat_visible(counts, gathered):
    """The global visible set as one ascending 1-D tensor (shards are ascending id ranges)."""
    c = counts.tolist()
    return torch.cat([gathered[r, :c[r]] for r in range(len(c))])


class VisibleExchange:
    """The path's one exchange, double-buffered: each rank's visibility mask travels in ONE fixed-size
    allgather on a side stream (clapgpu_exchange_visible: ncclAllGather from C, or torch.distributed as the
    fallback), and every rank expands the gathered mask into the identical ascending global id list
    (clapgpu_visible_compact over world * n entities).  Exchange + expansion of frame f overlap the update
    of frame f + 1: `begin()` hands the update kernel the mask buffer of this frame, `submit()` queues
    the exchange behind it.  Requires an initialised default process group."""

    def __init__(self, batch, rank, world, device, route="rccl", share=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.batch, self.rank, self.world, self.device = batch, rank, world, device
        self.n_pad = batch.n
        # shards need not be equal (clapgpu_shard_tile_range cuts whole tiles; a rank may be empty): every rank's padded
        # size, its first scene-global id (ascending ranges in rank order) and the common capacity the collective carries
        sizes = torch.zeros(world, dtype=torch.int64, device=device)
        sizes[rank] = self.n_pad
        dist.all_reduce(sizes)
        self.n_pad_all = np.ascontiguousarray(sizes.cpu().numpy(), np.uint32)
        self.base = np.ascontiguousarray(np.concatenate([[0], np.cumsum(self.n_pad_all.astype(np.int64))[:-1]]), np.uint32)
        self.cap_pad = max(64, int(self.n_pad_all.max()))
        if int(self.n_pad_all.astype(np.int64).sum()) > 0xffffffff:
            raise ValueError("the scene's global ids do not fit 32 bits")
        n_words = self.cap_pad // 64
        own = batch.vis_mask.numel()

        def mask_buf():                                      # cap_pad / 64 words, zero beyond this rank's own
            return torch.zeros(n_words, dtype=batch.vis_mask.dtype, device=device)
        self.masks = [mask_buf(), mask_buf()]
        self.masks[0][:own].copy_(batch.vis_mask)
        self.pops = [batch.vis_row_pop, torch.zeros_like(batch.vis_row_pop)]
        self.g_mask = [torch.zeros(world * n_words, dtype=torch.int64, device=device) for _ in range(2)]
        self.g_vis = [torch.zeros(world * self.cap_pad, dtype=torch.int32, device=device) for _ in range(2)]
        self.g_cnt = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(2)]
        nscratch = _lib.lib().clapgpu_visible_scratch_bytes(world * self.cap_pad)
        self.g_scratch = torch.zeros(nscratch // 4 + 4, dtype=torch.int32, device=device)
        self.comm = torch.cuda.Stream(device=device)
        self.ev_upd = [torch.cuda.Event() for _ in range(2)]
        self.ev_comm = [torch.cuda.Event() for _ in range(2)]
        self.frame = 0
        self.direct = None
        self.owns_direct = True
        if share is not None:                                # another workload of the same run: its communicator, its route
            self.direct, self.owns_direct = share.direct, False
        elif route == "rccl":
            import os
            import sys
            L = _lib.lib()
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(path):
                L.clapgpu_exchange_set_library(path.encode())
            # clapgpu_exchange_create is collective (ncclCommInitRank): ranks first agree that ALL of them can open RCCL
            # -- a rank that cannot would otherwise leave the others blocked inside the call -- and only then create it
            can = torch.tensor([int(L.clapgpu_exchange_available())], dtype=torch.int32, device=device)
            dist.all_reduce(can, op=dist.ReduceOp.MIN)
            if int(can.item()) == 1:
                try:
                    self.direct = self._create_exchange(rank, world, device)
                except Exception as exc:                    # keep the run alive: c10d does the same exchange
                    print(f"[clap_amd.shard] clapgpu_exchange unavailable ({exc}); using torch.distributed", file=sys.stderr)
                # and the same route afterwards: one rank falling back alone would deadlock the others
                ok = torch.tensor([1 if self.direct is not None else 0], dtype=torch.int32, device=device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and self.direct is not None:
                    L.clapgpu_exchange_destroy(self.direct)
                    self.direct = None
            elif rank == 0:
                print("[clap_amd.shard] RCCL cannot be opened on every rank; using torch.distributed", file=sys.stderr)

    def _create_exchange(self, rank, world, device):
        """libclapgpu's own exchange object (exchange.hip): RCCL opened at run time -- the copy torch has loaded --, the
        unique id carried to the ranks by the process group, once."""
        C, _lib = self._C, self._lib
        L = _lib.lib()
        uid = (C.c_uint8 * 128)()
        if rank == 0 and L.clapgpu_exchange_unique_id(uid) != 0:
            uid = (C.c_uint8 * 128)()                        # all zero = "no id": every rank then takes the fallback together
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
        dist.broadcast(t, src=0)                             # rank 0 always takes part, whatever happened above
        raw = (C.c_uint8 * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        if not any(raw):
            raise RuntimeError("rank 0 could not create an RCCL unique id: " + (L.clapgpu_last_error() or b"").decode())
        x = C.c_void_p()
        _lib.check(L.clapgpu_exchange_create(C.byref(x), raw, rank, world), "clapgpu_exchange_create")
        return x

    @property
    def route(self):
        return "ncclAllGather via clapgpu_exchange_visible (C)" if self.direct is not None else "torch.distributed all_gather"

    def begin(self):
        """Before the frame's update: wait until the exchange that last read this frame's mask buffer is done."""
        b = self.frame & 1
        torch.cuda.current_stream().wait_event(self.ev_comm[b])
        self.batch.use_vis_buffers(self.masks[b], self.pops[b])
        return b

    def submit(self):
        """After the frame's update was issued on the current stream."""
        C, _lib = self._C, self._lib
        b = self.frame & 1
        self.frame += 1
        main = torch.cuda.current_stream()
        self.ev_upd[b].record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ev_upd[b])
            if self.direct is not None:                    # allgather + expansion: one C call on the side stream
                rc = _lib.lib().clapgpu_exchange_visible_ranges(C.c_void_p(self.comm.cuda_stream), self.direct,
                                                                self.masks[b].data_ptr(), self.cap_pad, self.base.ctypes.data,
                                                                self.n_pad_all.ctypes.data, self.g_mask[b].data_ptr(),
                                                                self.g_vis[b].data_ptr(), self.g_cnt[b].data_ptr(),
                                                                self.g_scratch.data_ptr())
                _lib.check(rc, "clapgpu_exchange_visible_ranges")
            else:
                allgather_visible_mask(self.masks[b], self.world, self.g_mask[b])
                rc = _lib.lib().clapgpu_visible_compact_ranges(C.c_void_p(self.comm.cuda_stream), self.g_mask[b].data_ptr(),
                                                               self.world, self.cap_pad, self.base.ctypes.data,
                                                               self.n_pad_all.ctypes.data, self.g_vis[b].data_ptr(),
                                                               self.g_cnt[b].data_ptr(), self.g_scratch.data_ptr())
                _lib.check(rc, "clapgpu_visible_compact_ranges(global)")
            self.ev_comm[b].record(self.comm)

    def last(self):
        """(count tensor, id tensor) of the most recently submitted frame's global visible set; the caller
        synchronises (torch.cuda.synchronize or ev_comm) before reading."""
        b = (self.frame - 1) & 1
        return self.g_cnt[b], self.g_vis[b]

    def proof(self):
        """What the collective library itself says about the run, gathered over the ranks (collective: every rank calls it):
        dict(rccl_ranks = ncclCommCount of this rank's communicator or the process group's size on the c10d route,
        comm_rank_ok = ncclCommUserRank == rank on every rank, devices = PCI bus id of the device each rank drives, in rank
        order).  bench.py refuses to print a line when rccl_ranks != --gpus or two ranks share a device."""
        import ctypes as C
        L = self._lib.lib()
        ranks, me, bus = C.c_int(0), C.c_int(-1), C.create_string_buffer(32)
        if self.direct is not None and L.clapgpu_exchange_info(self.direct, C.byref(ranks), C.byref(me), bus) == 0:
            info = dict(rccl_ranks=int(ranks.value), comm_rank=int(me.value), device=bus.value.decode())
        else:                                                # torch.distributed route: the group's own size, torch's device record
            props = torch.cuda.get_device_properties(self.device)
            busid = getattr(props, "pci_bus_id", None)
            info = dict(rccl_ranks=dist.get_world_size(), comm_rank=dist.get_rank(),
                        device=f"{getattr(props, 'pci_domain_id', 0):04x}:{busid:02x}:{getattr(props, 'pci_device_id', 0):02x}.0"
                        if busid is not None else f"uuid:{getattr(props, 'uuid', self.device)}")
        every = [None] * self.world
        dist.all_gather_object(every, info)
        return dict(rccl_ranks=min(e["rccl_ranks"] for e in every),
                    comm_rank_ok=all(e["comm_rank"] == r for r, e in enumerate(every)),
                    devices=[e["device"] for e in every])

    def destroy(self):
        if self.direct is not None and self.owns_direct:
            self._lib.lib().clapgpu_exchange_destroy(self.direct)
        self.direct = None
