"""Host-side mirror of the reference's skeletal animation + skinning interface.

``SkinnedModel`` holds what ``model3d_add_skinning`` (model.c:524-538), ``animation_new`` /
``animation_add_channel`` (model.c:688-741) and ``model3d_make``'s vertex attributes set up
once; ``CharacterBatch.animated_update`` is the batched ``animated_update`` (model.c:1563-1592:
host time base -> channels_transform + one_joint_transform on the GPU) and
``CharacterBatch.skin`` the compute form of the skinning loop of shaders/model.vert:32-48.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else 0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(a, device, dtype=None):
    a = np.ascontiguousarray(a, dtype) if dtype is not None else np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(device)


def joint_depths(parent):
    """Level of every joint under joint 0; -1 for joints the reference's recursion from joint 0
    (model.c:1583, 1402-1403) never reaches."""
    parent = np.asarray(parent, np.int64)
    J = parent.shape[0]
    depth = np.full(J, -1, np.int32)
    if J:
        depth[0] = 0
    changed = True
    while changed:
        changed = False
        for j in range(1, J):
            p = parent[j]
            if depth[j] < 0 and p >= 0 and depth[p] >= 0:
                depth[j] = depth[p] + 1
                changed = True
    return depth


def channel_table(anims, nr_joints):
    """chan_table[a][joint][path] = (time_off, data_off, nr, 0) into the pooled key times / values
    of all animations; the last listed channel of a (joint, path) wins (model.c:1348-1349)."""
    table = np.zeros((len(anims), nr_joints, 3, 4), np.uint32)
    times, data = [], []
    t_base = d_base = 0
    for ai, an in enumerate(anims):
        for c in range(int(an["n_channels"])):
            tgt, path, nr = int(an["ch_target"][c]), int(an["ch_path"][c]), int(an["ch_nr"][c])
            if tgt < nr_joints and path < 3 and nr > 0:
                table[ai, tgt, path] = (t_base + int(an["ch_time_off"][c]), d_base + int(an["ch_data_off"][c]), nr, 0)
        times.append(an["times"])
        data.append(an["data"])
        t_base += an["times"].shape[0]
        d_base += an["data"].shape[0]
    return dict(chan_table=table, times=np.concatenate(times).astype(np.float32),
                data=np.concatenate(data).astype(np.float32))


def skeleton_bind(invmx):
    """model3d_add_skinning's bind = mat4x4_invert(invmx) per joint (model.c:524-537), with the library's arithmetic."""
    inv = np.ascontiguousarray(invmx, np.float32).reshape(-1, 16)
    out = np.zeros_like(inv)
    L, fp = _lib.lib(), C.POINTER(C.c_float)
    for j in range(inv.shape[0]):
        L.clapgpu_mat4_invert(inv[j].ctypes.data_as(fp), out[j].ctypes.data_as(fp))
    return out


class SkinnedModel:
    """Device copy of one model3d's skeleton, animations and (optionally) skinned mesh."""

    def __init__(self, sk, anims, mesh=None, bind=None, device="cuda:0"):
        self.device = dev = torch.device(device)
        self.nr_joints = J = int(sk["nr_joints"])
        self.depth_host = joint_depths(sk["parent"])
        self.n_levels = int(self.depth_host.max()) + 1
        self.parent = _dev(sk["parent"], dev, np.int32)
        self.depth = _dev(self.depth_host, dev, np.int32)
        self.root_pose = _dev(sk["root_pose"], dev, np.float32)
        self.invmx = _dev(sk["invmx"], dev, np.float32)
        if bind is None:
            bind = skeleton_bind(sk["invmx"])
        self.bind = _dev(bind, dev, np.float32)
        self.anims_host = anims
        ct = channel_table(anims, J)
        self.time_end = [float(a["time_end"]) for a in anims]
        self._ct = {k: _dev(v, dev) for k, v in ct.items()}
        self.skel_desc = _lib.Skeleton(J, self.n_levels, _ptr(self.parent), _ptr(self.depth),
                                       _ptr(self.root_pose), _ptr(self.invmx), _ptr(self.bind))
        self.anim_desc = _lib.Animations(len(anims), int(ct["times"].shape[0]), _ptr(self._ct["chan_table"]),
                                         _ptr(self._ct["times"]),
                                         _ptr(self._ct["data"]), None, 0, 0, int(ct["data"].shape[0]), 0)
        # the key-major copy of the pools with the rotation intervals' constants (clapgpu_animations_pack): once per
        # model, required by clapgpu_pose_update
        self.packed = None
        max_keys = int(ct["chan_table"][..., 2].max()) if len(anims) else 0
        max_keys = max(max_keys, 1)                       # animations without a single key: every path keeps its value
        if J <= 256 and len(anims):
            nbytes = int(_lib.lib().clapgpu_animations_packed_bytes(len(anims), max_keys, J))
            self.packed = torch.zeros((nbytes + 15) // 16 * 4, dtype=torch.float32, device=dev)
            layout = C.c_uint32(0)
            _lib.check(_lib.lib().clapgpu_animations_pack(_stream(), C.byref(self.anim_desc), J, max_keys, _ptr(self.packed),
                                                          C.byref(layout)), "clapgpu_animations_pack")
            self.anim_desc.packed = _ptr(self.packed)
            self.anim_desc.packed_keys = max_keys
            self.anim_desc.packed_layout = layout.value
        self.mesh = None
        if mesh is not None:
            self.mesh = dict(n_verts=int(mesh["n_verts"]), position=_dev(mesh["position"], dev, np.float32),
                             normal=_dev(mesh["normal"], dev, np.float32), joints=_dev(mesh["joints"], dev, np.uint8),
                             weights=_dev(mesh["weights"], dev, np.float32))


class CharacterBatch:
    """The animated entities of one SkinnedModel."""

    def __init__(self, model, n_chars, trs0, entity_mx, entity_index=None, vert_first=None, vert_count=None):
        self.model = model
        self.device = dev = model.device
        self.n = n = int(n_chars)
        J = model.nr_joints
        trs0 = np.asarray(trs0, np.float32)
        if trs0.ndim == 2:
            trs0 = np.broadcast_to(trs0, (n, J, 10))
        self.trs = _dev(trs0, dev, np.float32)
        self.entity_mx = entity_mx if torch.is_tensor(entity_mx) else _dev(entity_mx, dev, np.float32)
        self.entity_index = None if entity_index is None else _dev(entity_index, dev, np.uint32)
        self.anim = torch.zeros(n, dtype=torch.int32, device=dev)
        self.frame_time = torch.zeros(n, dtype=torch.float32, device=dev)
        self.joint_transforms = torch.zeros((n, J, 16), dtype=torch.float32, device=dev)
        self.joint_pos = torch.zeros((n, J, 4), dtype=torch.float32, device=dev)
        # host animation state of animated_update(): start time, speed, current animation
        self.ani_time = np.zeros(n, np.float64)
        self.speed = np.ones(n, np.float64)
        self.anim_host = np.zeros(n, np.int32)
        self._pose_desc = _lib.PoseBatch(n, 0, _ptr(self.anim), _ptr(self.frame_time), _ptr(self.entity_index),
                                         _ptr(self.entity_mx), _ptr(self.trs), _ptr(self.joint_transforms),
                                         _ptr(self.joint_pos))
        self._skin_desc = None
        if model.mesh is not None:
            if vert_first is None:                       # every character instances the whole mesh
                vert_first = np.zeros(n, np.uint32)
                vert_count = np.full(n, model.mesh["n_verts"], np.uint32)
            self.vert_first = _dev(vert_first, dev, np.uint32)
            self.vert_count_host = np.asarray(vert_count, np.uint32)
            self.vert_count = _dev(self.vert_count_host, dev, np.uint32)
            of = np.concatenate([[0], np.cumsum(self.vert_count_host.astype(np.int64))])
            self.out_first_host = of
            self.out_first = _dev(of[:-1], dev, np.uint32)
            total = int(of[-1])
            self.n_out_verts = total
            self.out_position = torch.zeros((total, 3), dtype=torch.float32, device=dev)
            self.out_normal = torch.zeros((total, 3), dtype=torch.float32, device=dev)
            m = model.mesh
            self._skin_desc = _lib.SkinBatch(n, J, _ptr(self.vert_first), _ptr(self.vert_count), _ptr(self.out_first),
                                             _ptr(m["position"]), _ptr(m["normal"]), _ptr(m["joints"]),
                                             _ptr(m["weights"]), _ptr(self.joint_transforms),
                                             _ptr(self.out_position), _ptr(self.out_normal), None)
            self.out_w = None

    # ---- animated_update (model.c:1563-1592) --------------------------------------------
    def set_frame_times(self, frame_time, anim=None):
        """Directly set each character's (float)frame_time (and animation id)."""
        self.frame_time.copy_(torch.from_numpy(np.ascontiguousarray(frame_time, np.float32)))
        if anim is not None:
            self.anim_host[:] = anim
            self.anim.copy_(torch.from_numpy(self.anim_host))

    def animated_update(self, now):
        """animated_update (model.c:1563-1592) for the batch, all on the device: the clock kernel turns
        (now, ani_time, speed) into the float frame times, flags `ended` and restarts repeating queue
        entries (animation_next -> animation_start: ani_time = now); then pose + palette.  The host
        only reads `ended` back when it has non-repeating entries or callbacks to serve."""
        if getattr(self, "_clock", None) is None:
            self.start_clock()
        if now is None:                                      # graph replay: the caller has written self.now_dev
            rc = _lib.lib().clapgpu_animation_time_dev(_stream(), C.byref(self._clock), _ptr(self.now_dev))
        else:
            rc = _lib.lib().clapgpu_animation_time(_stream(), C.byref(self._clock), float(now))
        _lib.check(rc, "clapgpu_animation_time")
        self.pose_update()

    def start_clock(self, ani_time=None, speed=None, repeat=None):
        """Device copies of entity3d.ani_time and the current queue entry's speed / repeat
        (model.h:417, model.c:1538-1541), from the host fields of the same names."""
        dev, n = self.device, self.n
        if ani_time is not None:
            self.ani_time[:] = ani_time
        if speed is not None:
            self.speed[:] = speed
        self.repeat_host = np.ones(n, np.uint8) if repeat is None else np.ascontiguousarray(repeat, np.uint8)
        self.ani_time_dev = torch.from_numpy(np.ascontiguousarray(self.ani_time, np.float64)).to(dev)
        self.speed_dev = torch.from_numpy(np.ascontiguousarray(self.speed, np.float32)).to(dev)
        self.repeat_dev = torch.from_numpy(self.repeat_host).to(dev)
        self.ended = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        self.now_dev = torch.zeros(1, dtype=torch.float64, device=dev)
        self.time_end_dev = torch.from_numpy(np.asarray(self.model.time_end, np.float32)).to(dev)
        self._clock = _lib.AnimClock(n, len(self.model.time_end), _ptr(self.anim), _ptr(self.time_end_dev),
                                     _ptr(self.ani_time_dev), _ptr(self.speed_dev), _ptr(self.repeat_dev),
                                     _ptr(self.frame_time), _ptr(self.ended))

    def download_clock(self):
        torch.cuda.synchronize(self.device)
        return dict(ani_time=self.ani_time_dev.cpu().numpy(), frame_time=self.frame_time.cpu().numpy(),
                    ended=self.ended.cpu().numpy()[:self.n])

    def set_outputs(self, trs=True, joint_pos=True):
        """Which host-visible by-products pose_update writes besides the palette (clapgpu_pose_batch.skip)."""
        self._pose_desc.skip = (0 if trs else _lib.POSE_SKIP_TRS) | (0 if joint_pos else _lib.POSE_SKIP_JOINT_POS)

    def set_joint_pos_model_space(self, on=True):
        """CLAPGPU_POSE_JOINT_POS_MODEL: pose_update leaves the model-space joint positions (model.c:1392-1397) in joint_pos
        and does not read the entity matrices; joint_pos_world() finishes model.c:1400."""
        if on:
            self._pose_desc.skip |= _lib.POSE_JOINT_POS_MODEL
        else:
            self._pose_desc.skip &= ~_lib.POSE_JOINT_POS_MODEL

    def joint_pos_world(self):
        skip = self._pose_desc.skip
        self._pose_desc.skip = skip & ~_lib.POSE_JOINT_POS_MODEL
        rc = _lib.lib().clapgpu_joint_pos_world(_stream(), C.byref(self.model.skel_desc), C.byref(self._pose_desc))
        self._pose_desc.skip = skip
        _lib.check(rc, "clapgpu_joint_pos_world")

    def pose_update(self):
        rc = _lib.lib().clapgpu_pose_update(_stream(), C.byref(self.model.skel_desc), C.byref(self.model.anim_desc),
                                            C.byref(self._pose_desc))
        _lib.check(rc, "clapgpu_pose_update")

    def set_skin_w(self, on=True):
        """Also write total_local_pos.w per vertex (clapgpu_skin_batch.out_w; model.vert:36-38,44)."""
        if on and self.out_w is None:
            self.out_w = torch.zeros(self.n_out_verts, dtype=torch.float32, device=self.device)
        self._skin_desc.out_w = _ptr(self.out_w) if on else None

    def skin(self):
        if self._skin_desc is None:
            raise ValueError("model has no skinned mesh")
        rc = _lib.lib().clapgpu_skin(_stream(), C.byref(self._skin_desc))
        _lib.check(rc, "clapgpu_skin")

    def download(self):
        torch.cuda.synchronize(self.device)
        out = dict(trs=self.trs.cpu().numpy(), joint_transforms=self.joint_transforms.cpu().numpy(),
                   joint_pos=self.joint_pos.cpu().numpy())
        if self._skin_desc is not None:
            out["out_position"] = self.out_position.cpu().numpy()
            out["out_normal"] = self.out_normal.cpu().numpy()
            if self.out_w is not None:
                out["out_w"] = self.out_w.cpu().numpy()
        return out

    def pose_algorithmic_bytes(self):
        return 200 * self.n * self.model.nr_joints

    def skin_algorithmic_bytes(self):
        return 68 * self.n_out_verts + 64 * self.model.nr_joints * self.n
