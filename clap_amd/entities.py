"""Host-side mirror of the reference's entity update / cull interface over device SoA.

Names follow the reference: ``mq_update`` (model.c:1953), ``view_entity_in_frustum``
(view.c:296), ``view_calc_frustum`` (view.c:291).  torch is used only for device
memory and streams; all arithmetic happens in libclapgpu's HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import synth


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def view_calc_frustum(cam):
    """view_update_from_angles + perspective + view_calc_frustum on the host
    (view.c:165-176, 178-193, 248-294).  cam: dict from synth.camera()."""
    L = _lib.lib()
    f32p = C.POINTER(C.c_float)
    pos = np.ascontiguousarray(cam["cam_pos"], np.float32)
    quat = np.ascontiguousarray(cam["cam_quat"], np.float32)
    view = np.zeros(16, np.float32)
    proj = np.zeros(16, np.float32)
    L.clapgpu_view_matrix(pos.ctypes.data_as(f32p), quat.ctypes.data_as(f32p), view.ctypes.data_as(f32p))
    fov, aspect, near, far = (float(v) for v in cam["persp"])
    z01 = int(cam["ndc_z_zero_one"][0])
    L.clapgpu_perspective(fov, aspect, near, far, z01, proj.ctypes.data_as(f32p))
    fr = _lib.Frustum()
    L.clapgpu_frustum_calc(view.ctypes.data_as(f32p), proj.ctypes.data_as(f32p), z01, C.byref(fr))
    return fr, view, proj


def frustum_arrays(fr):
    return (np.ctypeslib.as_array(fr.planes).reshape(6, 4).copy(),
            np.ctypeslib.as_array(fr.corners).reshape(8, 4).copy())


class EntityBatch:
    """Device-resident SoA of the entities whose update hook is default_update.

    Layout and padding rules: include/clapgpu.h (`clapgpu_entities`).  Build it from a
    level-padded scene dict (synth.pad_levels)."""

    IN_KEYS = ("pos_scale", "rot", "parent", "model", "flags", "seqs")

    def __init__(self, scene, device="cuda:0"):
        self.device = torch.device(device)
        dev = self.device
        self.n = int(scene["n"])
        self.n_real = int(scene.get("n_real", self.n))
        self.tiled = "tile_row_start" in scene
        if self.tiled:      # tile layout (clap_amd.tiler): one launch for all levels
            trs = np.ascontiguousarray(scene["tile_row_start"], np.uint32)
            if int(trs[-1]) * 64 != self.n:
                raise ValueError("tile_row_start does not cover the scene")
            self.n_tiles = len(trs) - 1
            self.tile_row_start = torch.from_numpy(trs.view(np.int32)).to(dev)
            self.level_start, self.n_levels = None, 0
        else:               # level-major layout (synth.pad_levels): one launch per level
            ls = np.asarray(scene["level_start"], np.uint32)
            if np.any(ls[:-1] % 64):
                raise ValueError("level starts must be multiples of 64: use synth.pad_levels()")
            self.level_start = np.ascontiguousarray(ls)
            self.n_levels = len(ls) - 1
            self._ls_ptr = self.level_start.ctypes.data_as(C.POINTER(C.c_uint32))
        for k in self.IN_KEYS:
            a = scene[k]
            if a.dtype == np.uint32:
                a = a.view(np.int32)
            setattr(self, k, torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        self.model_table = torch.from_numpy(synth.model_table(scene)).to(dev)
        n = self.n
        self.mx = torch.zeros((n, 16), dtype=torch.float32, device=dev)
        self.inv_mx = torch.zeros((n, 16), dtype=torch.float32, device=dev)
        self.aabb = torch.zeros((n, 6), dtype=torch.float32, device=dev)
        self.center = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        self.vis_mask = torch.zeros(((n + 63) // 64 or 1,), dtype=torch.int64, device=dev)
        n_rows = (n + 63) // 64
        self.vis_row_pop = torch.zeros(((n_rows + 15) // 16 * 16 or 16,), dtype=torch.uint8, device=dev)
        self.visible = torch.zeros((max(n, 1),), dtype=torch.int32, device=dev)
        self.visible_count = torch.zeros((1,), dtype=torch.int32, device=dev)
        nscratch = _lib.lib().clapgpu_visible_scratch_bytes(n)
        self.scratch = torch.zeros((nscratch // 4 or 1,), dtype=torch.int32, device=dev)
        self._desc = _lib.Entities(
            n=n, n_models=self.model_table.shape[0],
            pos_scale=self.pos_scale.data_ptr(), rot=self.rot.data_ptr(), parent=self.parent.data_ptr(),
            model=self.model.data_ptr(), model_table=self.model_table.data_ptr(), flags=self.flags.data_ptr(),
            seqs=self.seqs.data_ptr(), mx=self.mx.data_ptr(), inv_mx=self.inv_mx.data_ptr(),
            aabb=self.aabb.data_ptr(), center=self.center.data_ptr(), vis_mask=self.vis_mask.data_ptr(),
            vis_row_pop=self.vis_row_pop.data_ptr())

    def use_vis_buffers(self, vis_mask, vis_row_pop):
        """Point the kernels at another visibility mask / popcount pair (double buffering while a
        previous frame's mask is still being exchanged)."""
        self.vis_mask, self.vis_row_pop = vis_mask, vis_row_pop
        self._desc.vis_mask = vis_mask.data_ptr()
        self._desc.vis_row_pop = vis_row_pop.data_ptr()

    # ---- optional inputs ---------------------------------------------------------------
    def set_views(self, frusta):
        """The frame's other frusta (one light view for the shadow passes, pipeline-builder.c:246-272): culled by the same
        launch as the main one, each into its own mask plane (view_masks[v], view_row_pops[v]).  frusta: up to
        _lib.EXTRA_VIEWS_MAX _lib.Frustum; an empty list takes them off again."""
        dev = self.device
        n_rows = (self.n + 63) // 64
        self._views = _lib.Views()
        self._views.n = len(frusta)
        self.view_masks, self.view_row_pops = [], []
        for v, fr in enumerate(frusta[:_lib.EXTRA_VIEWS_MAX]):
            C.memmove(C.byref(self._views.frustum[v]), C.byref(fr), C.sizeof(_lib.Frustum))
            m = torch.zeros((n_rows or 1,), dtype=torch.int64, device=dev)
            p = torch.zeros(((n_rows + 15) // 16 * 16 or 16,), dtype=torch.uint8, device=dev)
            self.view_masks.append(m); self.view_row_pops.append(p)
            self._views.vis_mask[v] = m.data_ptr(); self._views.vis_row_pop[v] = p.data_ptr()
        self._desc.views = C.pointer(self._views) if frusta else None

    def compact_view(self, v, index_base=0):
        """The ascending visible list of extra view v (visible[:visible_count]), as compact_visible for the main one."""
        rc = _lib.lib().clapgpu_visible_compact(_stream(), _ptr(self.view_masks[v]), _ptr(self.view_row_pops[v]), self.n,
                                                index_base, _ptr(self.visible), _ptr(self.visible_count), _ptr(self.scratch))
        _lib.check(rc, "clapgpu_visible_compact")

    def set_attachments(self, attach, jt_pool, bind_pool):
        """Joint attachments (model.c:1626-1641).  attach: structured array (entity, jt, bind, pad)
        sorted by entity; jt_pool: device tensor of mat4 (e.g. CharacterBatch.joint_transforms);
        bind_pool: mat4 array.  Flags the entities CLAPGPU_E_JOINT_ATTACHED."""
        a = np.ascontiguousarray(attach)
        assert np.all(np.diff(a["entity"].astype(np.int64)) > 0), "attach table must be sorted by entity"
        self._attach = torch.from_numpy(a.view(np.uint8).reshape(-1, 16).copy()).to(self.device)
        self._jt_pool = jt_pool if torch.is_tensor(jt_pool) else torch.from_numpy(
            np.ascontiguousarray(jt_pool, np.float32)).to(self.device)
        self._bind_pool = torch.from_numpy(np.ascontiguousarray(bind_pool, np.float32)).to(self.device)
        idx = torch.from_numpy(a["entity"].astype(np.int64)).to(self.device)
        self.flags[idx] |= np.int32(_lib.E_JOINT_ATTACHED)
        self._desc.n_attach = a.shape[0]
        self._desc.attach = self._attach.data_ptr()
        self._desc.jt_pool = self._jt_pool.data_ptr()
        self._desc.bind_pool = self._bind_pool.data_ptr()
        self._attach_local = torch.zeros((a.shape[0], 16), dtype=torch.float32, device=self.device)
        self._desc.attach_local = self._attach_local.data_ptr()

    def set_bv_query(self, cam_pos, ctl_pos=None, ctl_entity=0):
        """Ask the next updates for default_update's camera bounding-volume pick."""
        self.bv_result = torch.zeros(1, dtype=torch.int64, device=self.device)
        q = _lib.BvQuery()
        q.cam_pos[:] = [float(v) for v in cam_pos]
        q.has_ctl = 0 if ctl_pos is None else 1
        q.ctl_pos[:] = [0.0, 0.0, 0.0] if ctl_pos is None else [float(v) for v in ctl_pos]
        q.ctl_entity = int(ctl_entity)
        q.result = self.bv_result.data_ptr()
        self._bv = q
        self._desc.bv = C.pointer(q)

    def camera_bv(self):
        """(entity index or -1, volume) of the last update's pick (host sync)."""
        key = int(self.bv_result.item()) & 0xFFFFFFFFFFFFFFFF
        if key == 0:
            return -1, 0.0
        return 0xFFFFFFFF - (key & 0xFFFFFFFF), float(np.asarray([key >> 32], np.uint32).view(np.float32)[0])

    # ---- reference-named operations -------------------------------------------------
    def mq_update(self, frustum=None, all_dirty=False):
        """mq_update over default_update entities; with `frustum` the cull of
        _models_render is fused into the same pass (vis_mask is written)."""
        mode = _lib.UPDATE_ALL_DIRTY if all_dirty else 0
        fr = C.byref(frustum) if frustum is not None else None
        if self.tiled:
            rc = _lib.lib().clapgpu_entities_update_tiles(_stream(), C.byref(self._desc), _ptr(self.tile_row_start),
                                                          self.n_tiles, mode, fr)
            _lib.check(rc, "clapgpu_entities_update_tiles")
        else:
            rc = _lib.lib().clapgpu_entities_update(_stream(), C.byref(self._desc), self._ls_ptr, self.n_levels,
                                                    mode, fr)
            _lib.check(rc, "clapgpu_entities_update")

    def update_level(self, level, frustum=None, all_dirty=False):
        """One hierarchy level of mq_update (callers that time or interleave per level)."""
        first = int(self.level_start[level])
        count = int(self.level_start[level + 1]) - first
        mode = _lib.UPDATE_ALL_DIRTY if all_dirty else 0
        rc = _lib.lib().clapgpu_entities_update_level(_stream(), C.byref(self._desc), first, count, mode,
                                                      C.byref(frustum) if frustum is not None else None)
        _lib.check(rc, "clapgpu_entities_update_level")

    def cull(self, frustum):
        """view_entity_in_frustum over every entity (one render pass)."""
        rc = _lib.lib().clapgpu_entities_cull(_stream(), C.byref(self._desc), C.byref(frustum))
        _lib.check(rc, "clapgpu_entities_cull")

    def compact_visible(self, index_base=0, two_pass=False):
        """Build the ascending visible-index list on the device (visible[:visible_count]).
        two_pass forces the large-n path (no per-row popcounts)."""
        rc = _lib.lib().clapgpu_visible_compact(_stream(), _ptr(self.vis_mask),
                                                None if two_pass else _ptr(self.vis_row_pop), self.n, index_base,
                                                _ptr(self.visible), _ptr(self.visible_count), _ptr(self.scratch))
        _lib.check(rc, "clapgpu_visible_compact")

    def alloc_lod(self):
        if not hasattr(self, "cur_lod"):
            self.cur_lod = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
            self.draw_lod = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
            self.force_lod = None

    def select_lod(self, cam_pos, force_lod=None):
        """LOD pick of the render pass for the compacted visible list (model.c:975-992):
        updates cur_lod, fills draw_lod[:visible_count] (the draw list is (visible, draw_lod))."""
        self.alloc_lod()
        if force_lod is not None:
            self.force_lod = torch.from_numpy(np.ascontiguousarray(force_lod, np.int32)).to(self.device)
        cp = (C.c_float * 3)(*[float(v) for v in cam_pos])
        rc = _lib.lib().clapgpu_entities_lod(_stream(), C.byref(self._desc), _ptr(self.visible),
                                             _ptr(self.visible_count), 0, cp, _ptr(self.force_lod),
                                             _ptr(self.cur_lod), _ptr(self.draw_lod))
        _lib.check(rc, "clapgpu_entities_lod")

    def compact_visible_lod(self, cam_pos, force_lod=None, index_base=0):
        """compact_visible() + select_lod() by one C call (clapgpu_visible_compact_lod): the render pass's ordered list and
        the LOD of every entry."""
        self.alloc_lod()
        if force_lod is not None:
            self.force_lod = torch.from_numpy(np.ascontiguousarray(force_lod, np.int32)).to(self.device)
        cp = (C.c_float * 3)(*[float(v) for v in cam_pos])
        rc = _lib.lib().clapgpu_visible_compact_lod(_stream(), C.byref(self._desc), index_base, cp, _ptr(self.force_lod),
                                                    _ptr(self.cur_lod), _ptr(self.visible), _ptr(self.visible_count),
                                                    _ptr(self.draw_lod), _ptr(self.scratch))
        _lib.check(rc, "clapgpu_visible_compact_lod")

    def view_entity_in_frustum(self, idx):
        """Served from the precomputed visibility bitmask (host sync)."""
        w = int(self.vis_mask[idx >> 6].item()) & 0xFFFFFFFFFFFFFFFF
        return bool((w >> (idx & 63)) & 1)

    # ---- host access ----------------------------------------------------------------
    def set_transforms(self, idx, pos_scale, rot):
        """entity3d_position/rotate/scale on a batch of entities: writes TRS, sets dirty."""
        idx_t = torch.as_tensor(np.asarray(idx, np.int64), device=self.device)
        self.pos_scale[idx_t] = torch.from_numpy(np.ascontiguousarray(pos_scale, np.float32)).to(self.device)
        self.rot[idx_t] = torch.from_numpy(np.ascontiguousarray(rot, np.float32)).to(self.device)
        self.flags[idx_t] |= np.int32(_lib.E_DIRTY)

    def download(self):
        torch.cuda.synchronize(self.device)
        cnt = int(self.visible_count.item())
        return dict(mx=self.mx.cpu().numpy(), inv_mx=self.inv_mx.cpu().numpy(), aabb=self.aabb.cpu().numpy(),
                    center=self.center.cpu().numpy(), flags=self.flags.cpu().numpy().view(np.uint32),
                    seqs=self.seqs.cpu().numpy().view(np.uint32),
                    vis_mask=self.vis_mask.cpu().numpy().view(np.uint64),
                    visible=self.visible[:cnt].cpu().numpy().view(np.uint32), visible_count=cnt)

    def algorithmic_bytes(self):
        """Algorithmic bytes of one all-dirty frame: 276 B per child, 212 B per root."""
        par = self.parent.cpu().numpy()
        real = (self.flags.cpu().numpy().view(np.uint32) & np.uint32(_lib.E_ALIVE)) != 0
        children = int(np.count_nonzero((par >= 0) & real))
        roots = int(np.count_nonzero((par < 0) & real))
        return 276 * children + 212 * roots
