"""clapgpu_entities_update_tiles_hostio: a host mirror's small frame as ONE launch (include/clapgpu.h) -- the touched
slots' inputs read from a device-mapped upload image, the rebuilt rows and the masks written to mapped result arrays, a
completion word raised -- against the plain clapgpu_entities_update_tiles on a second copy of the same scene and against
the oracle.  Replaces per entity: default_update / parent_transform_apply / entity3d_aabb_update
(core/model.c:1594-1695, 1200-1234) and view_entity_in_frustum (core/view.c:296-337)."""
import ctypes as C

import numpy as np
import pytest

from clap_amd import _lib, synth, tiler
from oracle import binding as ob

pytestmark = pytest.mark.gpu


class Hostio(C.Structure):
    _fields_ = [("pos_scale", C.c_void_p), ("rot", C.c_void_p), ("flags", C.c_void_p), ("touched", C.c_void_p),
                ("mx", C.c_void_p), ("inv_mx", C.c_void_p), ("aabb", C.c_void_p), ("center", C.c_void_p),
                ("vis_mask", C.c_void_p), ("rebuilt_mask", C.c_void_p), ("inside_mask", C.c_void_p),
                ("counter", C.c_void_p), ("done", C.c_void_p), ("done_value", C.c_uint32), ("options", C.c_uint32),
                ("keep_mask", C.c_void_p), ("exported_mask", C.c_void_p), ("stale_mask", C.c_void_p)]


class Mapped:
    """Page-locked host memory the device can address: numpy views on the host pointer, device alias for the kernel."""

    def __init__(self, nbytes):
        self.h, self.d = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().clapgpu_host_malloc_mapped(C.byref(self.h), C.byref(self.d), nbytes), "host_malloc_mapped")
        self.nbytes = nbytes
        self.bytes = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.h.value))
        self.bytes[:] = 0

    def view(self, off, count, dtype):
        return self.bytes[off:off + count * np.dtype(dtype).itemsize].view(dtype)

    def dev(self, off):
        return self.d.value + off

    def free(self):
        self.bytes = None
        _lib.lib().clapgpu_host_free(self.h)


def run_hostio(batch, io, fr, frame_id):
    import torch
    L = _lib.lib()
    io.done_value = frame_id
    rc = L.clapgpu_entities_update_tiles_hostio(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(batch._desc),
                                                C.c_void_p(batch.tile_row_start.data_ptr()), batch.n_tiles, 0,
                                                C.byref(fr) if fr is not None else None, C.byref(io))
    _lib.check(rc, "clapgpu_entities_update_tiles_hostio")


@pytest.mark.parametrize("cull", [True, False])
def test_one_launch_small_frame_matches_the_plain_update(cull, cuda_device):
    import torch
    from clap_amd import entities
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(23))
    raw = synth.entities_forest(6_000, 29, max_depth=7)
    scene = tiler.tiled_scene(raw)[0]
    n = int(scene["n"])
    words = n // 64
    cam = synth.camera(pos=(0, 0, 80))
    fr, _v, _p = entities.view_calc_frustum(cam)
    if not cull:
        fr = None
    a = entities.EntityBatch(scene, cuda_device)             # the plain path
    b = entities.EntityBatch(scene, cuda_device)             # the one-launch path
    a_rebuilt = torch.zeros(words + 2, dtype=torch.int64, device=a.device)
    b_rebuilt = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
    a._desc.rebuilt_mask = a_rebuilt.data_ptr()
    b._desc.rebuilt_mask = b_rebuilt.data_ptr()

    img = Mapped(n * 36 + (words + 2) * 8)                   # pos_scale | rot | flags | touched bits
    out = Mapped(n * 164 + 3 * (words + 2) * 8)              # mx | inv | aabb | center | vis | rebuilt | inside
    word = Mapped(64)
    counter = torch.zeros(1, dtype=torch.int32, device=b.device)
    h_ps, h_rot = img.view(0, 4 * n, np.float32).reshape(n, 4), img.view(16 * n, 4 * n, np.float32).reshape(n, 4)
    h_fl, h_touched = img.view(32 * n, n, np.uint32), img.view(36 * n, words + 2, np.uint64)
    h_ps[:], h_rot[:], h_fl[:] = scene["pos_scale"], scene["rot"], scene["flags"]
    o_mx, o_inv = out.view(0, 16 * n, np.float32).reshape(n, 16), out.view(64 * n, 16 * n, np.float32).reshape(n, 16)
    o_aabb, o_ctr = out.view(128 * n, 6 * n, np.float32).reshape(n, 6), out.view(152 * n, 3 * n, np.float32).reshape(n, 3)
    o_vis = out.view(164 * n, words + 2, np.uint64)
    o_reb = out.view(164 * n + (words + 2) * 8, words + 2, np.uint64)
    done = word.view(0, 1, np.uint32)
    io = Hostio(pos_scale=img.dev(0), rot=img.dev(16 * n), flags=img.dev(32 * n), touched=0,
                mx=out.dev(0), inv_mx=out.dev(64 * n), aabb=out.dev(128 * n), center=out.dev(152 * n),
                vis_mask=out.dev(164 * n), rebuilt_mask=out.dev(164 * n + (words + 2) * 8), inside_mask=0,
                counter=counter.data_ptr(), done=word.dev(0))
    try:
        mirror = dict(mx=np.zeros((n, 16), np.float32), inv=np.zeros((n, 16), np.float32),
                      aabb=np.zeros((n, 6), np.float32), ctr=np.zeros((n, 3), np.float32))
        for frame in range(5):
            a.mq_update(fr)
            io.touched = img.dev(36 * n) if frame else 0     # frame 0: the inputs went up with the batch, nothing flagged
            run_hostio(b, io, fr, frame + 1)
            _lib.check(L.clapgpu_wait_word(C.c_void_p(word.h.value), frame + 1, None), "clapgpu_wait_word")
            assert int(done[0]) == frame + 1
            # the completion word is the only synchronisation the host mirror uses: read the mapped arrays NOW
            reb = o_reb[:words].copy()
            bits = np.unpackbits(reb.view(np.uint8), bitorder="little").astype(bool)[:n]
            got = dict(mx=o_mx[bits].copy(), inv=o_inv[bits].copy(), aabb=o_aabb[bits].copy(), ctr=o_ctr[bits].copy(),
                       vis=o_vis[:words].copy())
            torch.cuda.synchronize()
            da, db = a.download(), b.download()
            what = f"frame {frame}, cull {cull}"
            for k in ("mx", "inv_mx", "aabb", "center", "flags", "seqs"):
                assert np.array_equal(da[k].view(np.uint32), db[k].view(np.uint32)), f"{what}: device {k} differs"
            assert np.array_equal(a_rebuilt.cpu().numpy()[:words].view(np.uint64), reb), f"{what}: rebuilt mask"
            assert np.array_equal(b_rebuilt.cpu().numpy()[:words].view(np.uint64), reb), f"{what}: rebuilt mask (device copy)"
            assert bits.any()
            has_box = np.asarray(scene["model_skip"])[scene["model"]] == 0
            assert np.array_equal(got["mx"].view(np.uint32), da["mx"][bits].view(np.uint32)), f"{what}: mapped mx"
            assert np.array_equal(got["inv"].view(np.uint32), da["inv_mx"][bits].view(np.uint32)), f"{what}: mapped inverse"
            hb = has_box[bits]
            assert np.array_equal(got["aabb"][hb].view(np.uint32), da["aabb"][bits][hb].view(np.uint32)), f"{what}: mapped aabb"
            assert np.array_equal(got["ctr"][hb].view(np.uint32), da["center"][bits][hb].view(np.uint32)), f"{what}: mapped center"
            if cull:
                assert np.array_equal(got["vis"], da["vis_mask"][:words]), f"{what}: mapped visibility mask"
            # a mirror that only ever copies the flagged rows stays equal to the device arrays
            mirror["mx"][bits], mirror["inv"][bits] = got["mx"], got["inv"]
            sel = np.flatnonzero(bits)[hb]
            mirror["aabb"][sel], mirror["ctr"][sel] = got["aabb"][hb], got["ctr"][hb]
            assert np.array_equal(mirror["mx"].view(np.uint32), da["mx"].view(np.uint32)), f"{what}: mirror drifted"
            assert np.array_equal(mirror["aabb"].view(np.uint32), da["aabb"].view(np.uint32)), f"{what}: mirror aabb drifted"
            # next frame: a tenth of the entities move; one in fifty only changes a flag (no DIRTY: entity3d_visible)
            h_touched[:] = 0
            alive = (scene["flags"] & np.uint32(_lib.E_ALIVE)) != 0
            move = (rng.uniform(0, 1, n) < 0.1) & alive
            hide = (rng.uniform(0, 1, n) < 0.02) & alive & ~move
            ps = h_ps.copy()
            ps[move, :3] += rng.uniform(-5, 5, (int(move.sum()), 3)).astype(np.float32)
            q = synth.quat_from_euler_xyz(*rng.uniform(-1, 1, (3, n))).astype(np.float32)
            rot = h_rot.copy()
            rot[move] = q[move]
            fl = db["flags"].copy()                           # what the device holds: DIRTY cleared where rebuilt
            fl[move] |= np.uint32(_lib.E_DIRTY)
            fl[hide] ^= np.uint32(_lib.E_VISIBLE)
            touched = move | hide
            h_ps[touched], h_rot[touched], h_fl[touched] = ps[touched], rot[touched], fl[touched]
            tb = np.packbits(touched, bitorder="little").view(np.uint64)
            h_touched[:len(tb)] = tb
            idx = np.flatnonzero(touched)
            a.pos_scale[torch.as_tensor(idx, device=a.device)] = torch.from_numpy(ps[idx]).to(a.device)
            a.rot[torch.as_tensor(idx, device=a.device)] = torch.from_numpy(rot[idx]).to(a.device)
            a.flags[torch.as_tensor(idx, device=a.device)] = torch.from_numpy(fl[idx].view(np.int32)).to(a.device)
            assert len(idx) > 100
    finally:
        torch.cuda.synchronize()
        img.free(); out.free(); word.free()


def test_one_launch_small_frame_matches_the_oracle(cuda_device):
    """The same entry against oracle/entity.c over three frames of touched subsets (bit for bit)."""
    import torch
    from clap_amd import entities
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(31))
    scene = tiler.tiled_scene(synth.entities_forest(3_000, 41, max_depth=6))[0]
    n, words = int(scene["n"]), int(scene["n"]) // 64
    cam = synth.camera(pos=(0, 0, 80))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    b = entities.EntityBatch(scene, cuda_device)
    reb_dev = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
    b._desc.rebuilt_mask = reb_dev.data_ptr()
    img = Mapped(n * 36 + (words + 2) * 8)
    out = Mapped(n * 164 + 3 * (words + 2) * 8)
    word = Mapped(64)
    counter = torch.zeros(1, dtype=torch.int32, device=b.device)
    h_ps, h_rot = img.view(0, 4 * n, np.float32).reshape(n, 4), img.view(16 * n, 4 * n, np.float32).reshape(n, 4)
    h_fl, h_touched = img.view(32 * n, n, np.uint32), img.view(36 * n, words + 2, np.uint64)
    h_ps[:], h_rot[:], h_fl[:] = scene["pos_scale"], scene["rot"], scene["flags"]
    o_mx = out.view(0, 16 * n, np.float32).reshape(n, 16)
    o_vis = out.view(164 * n, words + 2, np.uint64)
    o_reb = out.view(164 * n + (words + 2) * 8, words + 2, np.uint64)
    io = Hostio(pos_scale=img.dev(0), rot=img.dev(16 * n), flags=img.dev(32 * n), touched=img.dev(36 * n),
                mx=out.dev(0), inv_mx=out.dev(64 * n), aabb=out.dev(128 * n), center=out.dev(152 * n),
                vis_mask=out.dev(164 * n), rebuilt_mask=out.dev(164 * n + (words + 2) * 8), inside_mask=0,
                counter=counter.data_ptr(), done=word.dev(0))
    try:
        mirror = dict(mx=np.zeros((n, 16), np.float32), inv_mx=np.zeros((n, 16), np.float32))
        o_inv = out.view(64 * n, 16 * n, np.float32).reshape(n, 16)
        for frame in range(4):
            ob.entities_update(scene, st)
            vis, mask = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr_o)
            run_hostio(b, io, fr, frame + 1)
            _lib.check(L.clapgpu_wait_word(C.c_void_p(word.h.value), frame + 1, None), "clapgpu_wait_word")
            bits = np.unpackbits(o_reb[:words].copy().view(np.uint8), bitorder="little").astype(bool)[:n]
            mirror["mx"][bits], mirror["inv_mx"][bits] = o_mx[bits], o_inv[bits]
            for k in ("mx", "inv_mx"):
                assert np.array_equal(mirror[k].view(np.uint32), st[k].reshape(n, 16).view(np.uint32)), f"frame {frame}: {k}"
            assert np.array_equal(o_vis[:mask.size], mask), f"frame {frame}: visibility mask"
            torch.cuda.synchronize()
            db = b.download()
            assert np.array_equal(db["seqs"], st["seqs"]) and np.array_equal(db["flags"], st["flags"]), f"frame {frame}: seq / flags"
            assert np.array_equal(db["aabb"].view(np.uint32), st["aabb"].reshape(n, 6).view(np.uint32)), f"frame {frame}: aabb"
            h_touched[:] = 0
            alive = (scene["flags"] & np.uint32(_lib.E_ALIVE)) != 0
            move = (rng.uniform(0, 1, n) < 0.15) & alive
            ps = scene["pos_scale"].copy()
            ps[move, :3] += rng.uniform(-5, 5, (int(move.sum()), 3)).astype(np.float32)
            scene["pos_scale"][move] = ps[move]
            st["flags"][move] |= np.uint32(_lib.E_DIRTY)
            h_ps[move] = ps[move]
            h_fl[move] = st["flags"][move]
            tb = np.packbits(move, bitorder="little").view(np.uint64)
            h_touched[:len(tb)] = tb
    finally:
        torch.cuda.synchronize()
        img.free(); out.free(); word.free()


def test_hostio_argument_validation(cuda_device):
    from clap_amd import entities
    L = _lib.lib()
    scene = tiler.tiled_scene(synth.entities_forest(300, 3, max_depth=3))[0]
    b = entities.EntityBatch(scene, cuda_device)
    io = Hostio()
    rc = L.clapgpu_entities_update_tiles_hostio(None, C.byref(b._desc), C.c_void_p(b.tile_row_start.data_ptr()), b.n_tiles, 0,
                                                None, C.byref(io))
    assert rc == _lib.ERR_INVALID_ARGUMENTS
    rc = L.clapgpu_entities_update_tiles_hostio(None, C.byref(b._desc), C.c_void_p(b.tile_row_start.data_ptr()), b.n_tiles, 0,
                                                None, None)
    assert rc == _lib.ERR_INVALID_ARGUMENTS


class Export(C.Structure):
    _fields_ = [("mx", C.c_void_p), ("inv_mx", C.c_void_p), ("aabb", C.c_void_p), ("center", C.c_void_p),
                ("vis_mask", C.c_void_p), ("rebuilt_mask", C.c_void_p), ("inside_mask", C.c_void_p),
                ("counter", C.c_void_p), ("done", C.c_void_p), ("done_value", C.c_uint32), ("pad", C.c_uint32),
                ("stale_mask", C.c_void_p)]


@pytest.mark.parametrize("track", [False, True], ids=["plain", "stale-tracked"])
@pytest.mark.parametrize("cull", [True, False])
def test_export_policy_writes_back_what_is_read_and_the_rest_can_be_fetched(cull, track, cuda_device):
    """clapgpu_entities_hostio.keep_mask (what GPU_SCATTER_DRAWN rests on): with it the launch writes to the mapped result
    arrays exactly the rebuilt rows of entities that are drawn (vis_mask), contain a bounding-volume point, or are flagged
    in keep_mask -- exported_mask says which, every other mapped row is left as it was -- while the DEVICE arrays hold
    every rebuilt row as without the policy (against the oracle, bit for bit).  Without a frustum everything rebuilt comes
    back (a pass without a camera draws everything).  clapgpu_entities_export_rows then brings any selection over.
    stale-tracked: the launch also keeps clapgpu_entities_hostio.stale_mask on the device (rebuilt and not written = stale,
    written = not) and, with CLAPGPU_HOSTIO_EXPORT_STALE_READ, writes the stale rows that have a reader NOW although it did
    not rebuild them (the standing readers change every frame here): flagged in exported_mask, not in rebuilt_mask."""
    import torch
    from clap_amd import entities
    L = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(77))
    scene = tiler.tiled_scene(synth.entities_forest(5_000, 17, max_depth=6))[0]
    n, words = int(scene["n"]), int(scene["n"]) // 64
    cam = synth.camera(pos=(0, 5, 60))
    fr, _v, _p = entities.view_calc_frustum(cam)
    fr_o, _vo, _po = ob.frustum_from_camera(cam)
    if not cull:
        fr = None
    st = ob.entity_state(scene)
    b = entities.EntityBatch(scene, cuda_device)
    reb_dev = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
    inside_dev = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
    b._desc.rebuilt_mask = reb_dev.data_ptr()
    bvq = _lib.BvQuery()
    probe = scene["pos_scale"][np.flatnonzero(((scene["flags"] & np.uint32(_lib.E_ALIVE)) != 0) & (scene["parent"] < 0))[7], :3].copy()
    bvq.cam_pos[:] = [float(v) for v in probe]
    bvq.has_ctl, bvq.ctl_entity, bvq.result, bvq.inside_mask = 0, 0, None, inside_dev.data_ptr()
    b._desc.bv = C.pointer(bvq)
    img = Mapped(n * 36 + (words + 2) * 8)
    out = Mapped(n * 164 + 4 * (words + 2) * 8)
    word = Mapped(64)
    sel = Mapped((words + 2) * 8)
    counter = torch.zeros(1, dtype=torch.int32, device=b.device)
    keep = np.zeros(words + 2, np.uint64)
    alive = (scene["flags"] & np.uint32(_lib.E_ALIVE)) != 0
    kept = (rng.uniform(0, 1, n) < 0.05) & alive
    keep[:words] = np.packbits(kept, bitorder="little").view(np.uint64)
    keep_dev = torch.from_numpy(keep.view(np.int64)).to(b.device)
    h_ps, h_rot = img.view(0, 4 * n, np.float32).reshape(n, 4), img.view(16 * n, 4 * n, np.float32).reshape(n, 4)
    h_fl, h_touched = img.view(32 * n, n, np.uint32), img.view(36 * n, words + 2, np.uint64)
    h_ps[:], h_rot[:], h_fl[:] = scene["pos_scale"], scene["rot"], scene["flags"]
    mw = (words + 2) * 8
    o = dict(mx=out.view(0, 16 * n, np.float32).reshape(n, 16), inv=out.view(64 * n, 16 * n, np.float32).reshape(n, 16),
             aabb=out.view(128 * n, 6 * n, np.float32).reshape(n, 6), ctr=out.view(152 * n, 3 * n, np.float32).reshape(n, 3))
    o_vis, o_reb = out.view(164 * n, words + 2, np.uint64), out.view(164 * n + mw, words + 2, np.uint64)
    o_ins, o_exp = out.view(164 * n + 2 * mw, words + 2, np.uint64), out.view(164 * n + 3 * mw, words + 2, np.uint64)
    io = Hostio(pos_scale=img.dev(0), rot=img.dev(16 * n), flags=img.dev(32 * n), touched=img.dev(36 * n),
                mx=out.dev(0), inv_mx=out.dev(64 * n), aabb=out.dev(128 * n), center=out.dev(152 * n),
                vis_mask=out.dev(164 * n), rebuilt_mask=out.dev(164 * n + mw), inside_mask=out.dev(164 * n + 2 * mw),
                counter=counter.data_ptr(), done=word.dev(0), keep_mask=keep_dev.data_ptr(), exported_mask=out.dev(164 * n + 3 * mw))
    stale_dev = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
    if track:
        io.stale_mask, io.options = stale_dev.data_ptr(), 1
    POISON = np.float32(-12345.5)
    bits = lambda m: np.unpackbits(np.ascontiguousarray(m[:words]).view(np.uint8), bitorder="little").astype(bool)[:n]
    has_box = np.asarray(scene["model_skip"])[scene["model"]] == 0
    try:
        stale = np.zeros(n, bool)
        for frame in range(4):
            ob.entities_update(scene, st)
            vis, mask = ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr_o)
            for a in o.values():
                a[:] = POISON
            run_hostio(b, io, fr, frame + 1)
            _lib.check(L.clapgpu_wait_word(C.c_void_p(word.h.value), frame + 1, None), "clapgpu_wait_word")
            torch.cuda.synchronize()
            reb, exp, ins = bits(o_reb), bits(o_exp), bits(o_ins)
            db = b.download()
            for k, ok in (("mx", "mx"), ("inv_mx", "inv_mx")):
                assert np.array_equal(db[k].view(np.uint32), st[ok].reshape(n, 16).view(np.uint32)), f"frame {frame}: device {k}"
            visb = bits(o_vis) if cull else np.ones(n, bool)
            if cull:
                assert np.array_equal(o_vis[:mask.size], mask)
            box = st["aabb"].reshape(n, 6)
            inside_ref = alive & np.all(probe >= box[:, :3], axis=1) & np.all(probe <= box[:, 3:], axis=1)
            assert np.array_equal(ins, inside_ref), f"frame {frame}: containment mask"
            read_now = (visb | ins | kept)
            late = (stale & read_now & ~reb & alive) if track else np.zeros(n, bool)
            want = (reb & read_now) | late
            assert np.array_equal(exp, want), f"frame {frame}: exported = rebuilt & (drawn | containing | kept) [+ stale rows read now]"
            assert reb.sum() > 50 and (cull and 0 < (exp & reb).sum() < reb.sum() or not cull and (exp & reb).sum() == reb.sum())
            if track and cull and frame:
                assert late.sum() > 0, "the scenario has no stale row with a new reader"
            if track:
                assert np.array_equal(bits(stale_dev.cpu().numpy().view(np.uint64)), (stale | reb) & ~exp), f"frame {frame}: device stale mask"
            assert np.array_equal(o["mx"][exp].view(np.uint32), db["mx"][exp].view(np.uint32)), f"frame {frame}: exported mx"
            assert np.array_equal(o["inv"][exp].view(np.uint32), db["inv_mx"][exp].view(np.uint32))
            eb = exp & has_box
            assert np.array_equal(o["aabb"][eb].view(np.uint32), db["aabb"][eb].view(np.uint32))
            assert np.array_equal(o["ctr"][eb].view(np.uint32), db["center"][eb].view(np.uint32))
            assert (o["mx"][~exp] == POISON).all() and (o["aabb"][~exp] == POISON).all(), f"frame {frame}: a row nobody reads was written"
            stale = (stale | reb) & ~exp
            # fetch a third of what is stale, by mask
            pick = stale & (rng.uniform(0, 1, n) < 0.34)
            sel.view(0, words + 2, np.uint64)[:words] = np.packbits(pick, bitorder="little").view(np.uint64)
            x = Export(mx=out.dev(0), inv_mx=out.dev(64 * n), aabb=out.dev(128 * n), center=out.dev(152 * n),
                       counter=counter.data_ptr(), done=word.dev(0), done_value=1000 + frame,
                       stale_mask=stale_dev.data_ptr() if track else None)
            _lib.check(L.clapgpu_entities_export_rows(None, C.byref(b._desc), C.byref(x), C.c_void_p(sel.dev(0))), "export_rows")
            _lib.check(L.clapgpu_wait_word(C.c_void_p(word.h.value), 1000 + frame, None), "clapgpu_wait_word")
            assert np.array_equal(o["mx"][pick].view(np.uint32), db["mx"][pick].view(np.uint32)), f"frame {frame}: fetched rows"
            assert np.array_equal(o["aabb"][pick & has_box].view(np.uint32), db["aabb"][pick & has_box].view(np.uint32))
            assert (o["mx"][~exp & ~pick] == POISON).all(), f"frame {frame}: the fetch wrote a row it was not asked for"
            assert np.array_equal(bits(o_reb), reb) and np.array_equal(bits(o_exp), exp), "a fetch leaves the frame's masks alone"
            stale &= ~pick
            if track:
                torch.cuda.synchronize()
                assert np.array_equal(bits(stale_dev.cpu().numpy().view(np.uint64)), stale), f"frame {frame}: a fetch clears the stale bits of what it hands over"
                kept = (rng.uniform(0, 1, n) < 0.05) & alive            # other standing readers next frame
                keep[:words] = np.packbits(kept, bitorder="little").view(np.uint64)
                keep_dev.copy_(torch.from_numpy(keep.view(np.int64)))
            # next frame: a fifth of the entities move
            h_touched[:] = 0
            move = (rng.uniform(0, 1, n) < 0.2) & alive
            ps = scene["pos_scale"].copy()
            ps[move, :3] += rng.uniform(-5, 5, (int(move.sum()), 3)).astype(np.float32)
            scene["pos_scale"][move] = ps[move]
            st["flags"][move] |= np.uint32(_lib.E_DIRTY)
            h_ps[move] = ps[move]
            h_fl[move] = st["flags"][move]
            h_touched[:words] = np.packbits(move, bitorder="little").view(np.uint64)
    finally:
        torch.cuda.synchronize()
        b._desc.bv = None
        img.free(); out.free(); word.free(); sel.free()
