"""One display frame of the batched path, in the reference's order.

``clap_frame()`` (core/clap.c:551-665) runs: ``phys_step`` (clap.c:604) -> ``scene_update`` ->
``mq_update`` with every entity's update hook in list order -- ``character_update`` (character.c:583)
in front of ``default_update`` (model.c:1649: body read-back, TRS rebuild, light hand-off, rotation
push to colliders, ``animated_update``), ``particles_update`` (particle.c:89) -> camera / light grid
-> render passes (frustum test, LOD pick per drawn entity; the vertex shader skins).  The sequence itself
lives in C: ``clapgpu_frame_issue`` (clap_amd/csrc/frame.hip) issues it as a fixed series of launches on one
stream from a ``clapgpu_frame`` descriptor; ``FrameLoop`` only gathers the descriptor from the harness objects,
keeps the host-side time base and captures / replays the call as a HIP graph.  Nothing is read back unless asked.
(A variant that forked the particle systems and one broadphase pass onto a side stream was measured in round 1:
no faster at BASELINE size, slower at testbed size; removed.)
"""
import numpy as np
import torch

from . import entities as ent_mod


class FrameLoop:
    def __init__(self, batch, cam, world=None, feed=None, body_links=None, lights=None, characters=None,
                 particles=None, contacts=False, pose_readers=("trs", "joint_pos"), prebin=False):
        """batch: EntityBatch.  world: PhysWorld (dynamic bodies write their entities through
        body_entity; character bodies have body_entity = -1).  feed: CharacterFeed.  body_links:
        (link_body, link_entity) of characters / static colliders whose rotation follows the entity.
        lights: LightSet (with carriers).  characters: CharacterBatch (pose + skin).  particles:
        ParticleBatch.  pose_readers: which of the pose's host-visible by-products somebody reads this frame --
        "trs" (struct joint's translation / rotation / scale) and "joint_pos" (struct joint.pos: camera_target,
        camera.c:191-205); the draw path and the skinning consume joint_transforms alone (model.c:1020-1022), so a frame
        whose skinning runs on the device registers none and the pose writes 64 of its 120 bytes per joint
        (clapgpu_pose_batch.skip).  A model with (joint, path) pairs that have no channel keeps "trs": such a path's
        value lives there (model.c:1301)."""
        self.batch, self.world, self.feed, self.lights = batch, world, feed, lights
        self.characters, self.particles = characters, particles
        self.body_links, self.contacts = body_links, contacts
        self.prebin = prebin        # CLAPGPU_FRAME_PREBIN: the step bins its boxes for the next frame's broadphase (nothing else writes them)
        self._desc = None
        if characters is not None:
            missing = bool(characters.model.anim_desc.packed_layout & 0x010)        # POSE_LAYOUT_MISSING (pose.hip)
            characters.set_outputs(trs=("trs" in pose_readers) or missing, joint_pos="joint_pos" in pose_readers)
        self.set_camera(cam)

    def set_camera(self, cam):
        self.cam = cam
        self.frustum, self.view_mx, self.proj_mx = ent_mod.view_calc_frustum(cam)
        self._desc = None                                   # frustum / matrices are part of the descriptor

    # ---- the clapgpu_frame descriptor ---------------------------------------------------
    def _build(self):
        import ctypes as C
        from . import _lib
        b, w = self.batch, self.world
        f = _lib.Frame()
        keep = []                                           # ctypes objects the descriptor points into
        f.entities = C.pointer(b._desc)
        if b.tiled:
            f.tile_row_start, f.n_tiles = b.tile_row_start.data_ptr(), b.n_tiles
        else:
            f.level_start, f.n_levels = b.level_start.ctypes.data, b.n_levels
        f.frustum = C.pointer(self.frustum)
        if w is not None:
            f.bodies, f.world, f.bp = C.pointer(w._desc), C.pointer(w.world), w._bp
            f.pairs, f.pair_capacity, f.pair_total = w.pairs.data_ptr(), w.capacity, w.pair_total.data_ptr()
            if w.n_static:
                f.static_pairs, f.static_pair_capacity = w.static_pairs.data_ptr(), w.static_capacity
                f.static_pair_total = w.static_pair_total.data_ptr()
            if self.contacts:
                w.alloc_contacts()
                g = w.body_geoms()
                keep.append(g)
                f.body_geoms = C.pointer(g)
                f.contacts, f.contact_total = w.contact2_buf.data_ptr(), w.contact2_total.data_ptr()
                if w.n_static:
                    sg = w.static_geoms()
                    keep.append(sg)
                    f.static_geoms = C.pointer(sg)
                    f.static_contacts = w.static_contact2_buf.data_ptr()
                    f.static_contact_total = w.static_contact2_total.data_ptr()
            if self.body_links is not None:
                lb, le = w.upload_links(*self.body_links)
                f.n_body_links, f.link_body, f.link_entity = len(self.body_links[0]), lb.data_ptr(), le.data_ptr()
        if self.feed is not None:
            f.characters = C.pointer(self.feed._desc)
        vm = np.ascontiguousarray(self.view_mx, np.float32)
        pm = np.ascontiguousarray(self.proj_mx, np.float32)
        keep += [vm, pm]
        f.view_mx = vm.ctypes.data_as(C.POINTER(C.c_float))
        f.proj_mx = pm.ctypes.data_as(C.POINTER(C.c_float))
        ls = self.lights
        if ls is not None:
            d = ls._desc()
            keep.append(d)
            f.lights = C.pointer(d)
            if ls._carriers and ls._carriers[0]:
                n, ce, cl, co = ls._carriers
                f.n_light_carriers, f.carrier_entity, f.carrier_light, f.carrier_offset = n, ce.data_ptr(), cl.data_ptr(), co.data_ptr()
            tiles = ls.alloc_tiles()
            if tiles is not None:
                f.light_width, f.light_height, f.light_cell, f.light_tiles = ls.width, ls.height, ls.cell, tiles.data_ptr()
        cb = self.characters
        if cb is not None:
            if getattr(cb, "_clock", None) is None:
                cb.start_clock()
            f.anim_clock = C.pointer(cb._clock)
            f.skeleton, f.animations = C.pointer(cb.model.skel_desc), C.pointer(cb.model.anim_desc)
            f.pose = C.pointer(cb._pose_desc)
            if cb._skin_desc is not None:
                f.skin = C.pointer(cb._skin_desc)
        if self.particles is not None:
            f.particles = C.pointer(self.particles._desc)
        b.alloc_lod()
        f.index_base = 0
        f.visible, f.visible_count, f.visible_scratch = b.visible.data_ptr(), b.visible_count.data_ptr(), b.scratch.data_ptr()
        f.cam_pos[:] = [float(v) for v in self.cam["cam_pos"]]
        f.force_lod = b.force_lod.data_ptr() if b.force_lod is not None else None
        f.cur_lod, f.draw_lod = b.cur_lod.data_ptr(), b.draw_lod.data_ptr()
        self._desc, self._keep = f, keep
        return f

    def capture(self, dt=1.0 / 120.0, warmup_now=0.0):
        """Record one frame (with one physics substep, the current camera) as a HIP graph.  Afterwards
        clap_frame_replay(now) costs one graph launch instead of ~35 kernel launches: what a
        testbed-sized scene, whose frame is launch latency, needs.  Re-capture after a camera change
        (the frustum and view matrices are kernel arguments)."""
        self.clap_frame(warmup_now, dt)                      # everything allocated, lazily created state exists
        torch.cuda.synchronize()
        self._now_host = torch.zeros(1, dtype=torch.float64).pin_memory()
        self._graph = torch.cuda.CUDAGraph()
        saved = (None if self.world is None else self.world.time_acc.value)
        with torch.cuda.graph(self._graph):
            self._issue(None, steps=1)
        if self.world is not None:
            self.world.time_acc.value = saved

    def clap_frame_replay(self, now):
        self._now_host[0] = float(now)
        if self.characters is not None:
            self.characters.now_dev.copy_(self._now_host, non_blocking=True)
        self._graph.replay()

    def clap_frame(self, now, dt):
        steps = self.world.phys_step_begin(dt) if self.world is not None else 0
        self._issue(now, steps)

    def _issue(self, now, steps):
        import ctypes as C
        from . import _lib
        f = self._desc or self._build()
        if self.lights is not None:
            self.lights._upload()                            # slot edits made on the host since the last frame
        # graph capture: the clock comes from a device double written before every replay
        f.now_dev = self.characters.now_dev.data_ptr() if (now is None and self.characters is not None) else None
        # CLAPGPU_FRAME_OVERLAP: three chains on three streams (frame.hip); CLAPGPU_FRAME_PREBIN
        f.flags = (1 if getattr(self, "overlap", False) else 0) | (2 if self.prebin else 0)
        rc = _lib.lib().clapgpu_frame_issue(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(f),
                                            0.0 if now is None else float(now), int(steps))
        _lib.check(rc, "clapgpu_frame_issue")
