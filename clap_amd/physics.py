"""Host-side mirror of the reference's physics interface for free sphere bodies.

``PhysWorld.phys_step(dt)`` follows phys_step() / __phys_step() (physics.c:746-787): the
fixed-step schedule runs on the host, each substep does the two broadphase calls and the ODE
world step on the GPU; ``phys_body_update`` is the body -> entity read-back of
physics.c:789-812.  ODE being absent from the reference tree, this block is parity-unpinned
(DESIGN.md).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else 0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class PhysWorld:
    def __init__(self, bodies, statics=None, pair_capacity=None, device="cuda:0"):
        """bodies: dict from synth.sphere_bodies(); statics: float64 [ns, 6] (minx,maxx,miny,maxy,minz,maxz)."""
        self.device = dev = torch.device(device)
        self.n = n = int(bodies["n"])
        t = lambda k, dt: torch.from_numpy(np.ascontiguousarray(bodies[k], dt)).to(dev)
        self.pos, self.quat = t("pos", np.float64), t("quat", np.float64)
        self.lvel, self.avel = t("lvel", np.float64), t("avel", np.float64)
        self.mass, self.radius, self.yoffset = t("mass", np.float64), t("radius", np.float64), t("yoffset", np.float64)
        self.bflags = torch.from_numpy(np.ascontiguousarray(bodies["bflags"]).view(np.int32)).to(dev)
        self.adis_steps_left = t("adis_steps_left", np.int32)
        self.adis_time_left = t("adis_time_left", np.float64)
        self.body_entity = t("body_entity", np.int32)
        self.cell = float(bodies["cell"])
        self.world = _lib.World()
        _lib.lib().clapgpu_world_defaults(C.byref(self.world))
        self.time_acc = C.c_double(0.0)
        self._desc = _lib.Bodies(n, 0, _ptr(self.pos), _ptr(self.quat), _ptr(self.lvel), _ptr(self.avel),
                                 _ptr(self.mass), _ptr(self.radius), _ptr(self.yoffset), _ptr(self.bflags),
                                 _ptr(self.adis_steps_left), _ptr(self.adis_time_left), _ptr(self.body_entity))
        self.capacity = int(pair_capacity if pair_capacity is not None else max(8 * n, 1024))
        self.pairs = torch.zeros((self.capacity, 2), dtype=torch.int32, device=dev)
        self.pair_total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.static_pairs = torch.zeros((self.capacity, 2), dtype=torch.int32, device=dev)
        self.static_pair_total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(_lib.lib().clapgpu_broadphase_scratch_bytes(n) // 4 + 4, dtype=torch.int32, device=dev)
        self.statics = None
        self.n_static = 0
        if statics is not None and len(statics):
            self.statics = torch.from_numpy(np.ascontiguousarray(statics, np.float64)).to(dev)
            self.n_static = self.statics.shape[0]

    # ---- __phys_step pieces -----------------------------------------------------------
    def broadphase(self, side=None):
        """dSpaceCollide2(ground, bodies) + dSpaceCollide(bodies): candidate pair lists.

        side: a second stream.  The two passes read the same body state and write disjoint outputs, and both
        are chains of small latency-bound launches, so with `side` the statics pass runs there (on its own
        scratch) while the bodies pass runs on the current stream; world_step() joins before it moves bodies."""
        L = _lib.lib()
        if self.n_static and side is not None:
            if getattr(self, "scratch_static", None) is None:
                self.scratch_static = torch.zeros_like(self.scratch)
            side.wait_stream(torch.cuda.current_stream())        # the bodies are where the last step left them
            with torch.cuda.stream(side):
                _lib.check(L.clapgpu_broadphase_static_pairs(_stream(), C.byref(self._desc), self.n_static,
                                                             _ptr(self.statics), _ptr(self.static_pairs), self.capacity,
                                                             _ptr(self.static_pair_total), _ptr(self.scratch_static)),
                           "clapgpu_broadphase_static_pairs")
                self._static_done = side.record_event()
        elif self.n_static:
            _lib.check(L.clapgpu_broadphase_static_pairs(_stream(), C.byref(self._desc), self.n_static,
                                                         _ptr(self.statics), _ptr(self.static_pairs), self.capacity,
                                                         _ptr(self.static_pair_total), _ptr(self.scratch)),
                       "clapgpu_broadphase_static_pairs")
        _lib.check(L.clapgpu_broadphase_pairs(_stream(), C.byref(self._desc), self.cell, _ptr(self.pairs),
                                              self.capacity, _ptr(self.pair_total), _ptr(self.scratch)),
                   "clapgpu_broadphase_pairs")

    def rotate_from_entities(self, entity_batch, link_body, link_entity, all_dirty=False):
        """phys_body_rotate_xform for the (body, entity) links whose entity default_update is about to
        rebuild (model.c:1680-1687); run before entity_batch.mq_update."""
        key = (id(link_body), id(link_entity))
        if getattr(self, "_links_key", None) != key:         # uploaded once per link table (also keeps graph capture clean)
            self._links_key = key
            self._links = (torch.from_numpy(np.ascontiguousarray(link_body, np.uint32).view(np.int32)).to(self.device),
                           torch.from_numpy(np.ascontiguousarray(link_entity, np.uint32).view(np.int32)).to(self.device))
        lb, le = self._links
        rc = _lib.lib().clapgpu_bodies_rotate_from_entities(_stream(), C.byref(self._desc), C.byref(entity_batch._desc),
                                                            _lib.UPDATE_ALL_DIRTY if all_dirty else 0, len(link_body),
                                                            _ptr(lb), _ptr(le))
        _lib.check(rc, "clapgpu_bodies_rotate_from_entities")

    def set_materials(self, material):
        """Per-body phys_body parameters (bounce, bounce_vel, mu, soft_erp, soft_cfm; physics.c:77-81)."""
        self.material = torch.from_numpy(np.ascontiguousarray(material, np.float64)).to(self.device)

    def contacts(self):
        """near_callback on the body x body candidate pairs of the last broadphase(): one contact record
        per pair (oracle.binding.CONTACT_DTYPE layout = clapgpu_contact) + the number of touching pairs."""
        if getattr(self, "contact_buf", None) is None:
            self.contact_buf = torch.zeros((self.capacity, 104), dtype=torch.uint8, device=self.device)
            self.contact_total = torch.zeros(1, dtype=torch.int32, device=self.device)
        mat = getattr(self, "material", None)
        _lib.check(_lib.lib().clapgpu_contacts_spheres(_stream(), C.byref(self._desc), _ptr(self.pairs),
                                                       _ptr(self.pair_total), self.capacity, _ptr(mat),
                                                       _ptr(self.contact_buf), _ptr(self.contact_total)),
                   "clapgpu_contacts_spheres")

    def contacts_static(self, static_material=None):
        """near_callback on the (body, static box) candidate pairs of the last broadphase(): ODE's
        dCollideSphereBox + phys_contact_surface, one record per pair."""
        if getattr(self, "static_contact_buf", None) is None:
            self.static_contact_buf = torch.zeros((self.capacity, 104), dtype=torch.uint8, device=self.device)
            self.static_contact_total = torch.zeros(1, dtype=torch.int32, device=self.device)
        if static_material is not None:
            self.static_material = torch.from_numpy(np.ascontiguousarray(static_material, np.float64)).to(self.device)
        mat, smat = getattr(self, "material", None), getattr(self, "static_material", None)
        _lib.check(_lib.lib().clapgpu_contacts_sphere_box(_stream(), C.byref(self._desc), self.n_static,
                                                          _ptr(self.statics), _ptr(self.static_pairs),
                                                          _ptr(self.static_pair_total), self.capacity, _ptr(mat),
                                                          _ptr(smat), _ptr(self.static_contact_buf),
                                                          _ptr(self.static_contact_total)),
                   "clapgpu_contacts_sphere_box")

    def download_static_contacts(self, dtype):
        torch.cuda.synchronize(self.device)
        npairs = min(int(self.static_pair_total.item()), self.capacity)
        return (self.static_contact_buf[:npairs].cpu().numpy().view(dtype).reshape(-1),
                int(self.static_contact_total.item()))

    def download_contacts(self, dtype):
        torch.cuda.synchronize(self.device)
        npairs = min(int(self.pair_total.item()), self.capacity)
        return self.contact_buf[:npairs].cpu().numpy().view(dtype).reshape(-1), int(self.contact_total.item())

    def world_step(self, h):
        ev = getattr(self, "_static_done", None)
        if ev is not None:                                   # the statics pass on the side stream still reads pos
            torch.cuda.current_stream().wait_event(ev)
            self._static_done = None
        _lib.check(_lib.lib().clapgpu_bodies_step(_stream(), C.byref(self._desc), C.byref(self.world), h),
                   "clapgpu_bodies_step")

    def phys_step_begin(self, dt):
        """The schedule half of phys_step (physics.c:773-787): number of fixed substeps for this frame."""
        return _lib.lib().clapgpu_phys_step_schedule(C.byref(self.time_acc), dt)

    def phys_step(self, dt, broadphase=True):
        """phys_step(phys, dt): returns the number of fixed substeps taken."""
        steps = _lib.lib().clapgpu_phys_step_schedule(C.byref(self.time_acc), dt)
        for _ in range(steps):
            if broadphase:
                self.broadphase()
            self.world_step(1.0 / 120.0)
        return steps

    def phys_body_update(self, entity_batch, moving=None):
        """phys_body_update for every body: entity TRS <- body pose, entities marked dirty."""
        _lib.check(_lib.lib().clapgpu_phys_body_update(_stream(), C.byref(self._desc), entity_batch.n,
                                                       _ptr(entity_batch.pos_scale),
                                                       _ptr(entity_batch.rot), _ptr(entity_batch.flags),
                                                       _ptr(moving)),
                   "clapgpu_phys_body_update")

    def download(self):
        torch.cuda.synchronize(self.device)
        npairs = int(self.pair_total.item())
        nst = int(self.static_pair_total.item())
        return dict(pos=self.pos.cpu().numpy(), quat=self.quat.cpu().numpy(), lvel=self.lvel.cpu().numpy(),
                    avel=self.avel.cpu().numpy(), bflags=self.bflags.cpu().numpy().view(np.uint32),
                    adis_steps_left=self.adis_steps_left.cpu().numpy(), adis_time_left=self.adis_time_left.cpu().numpy(),
                    pair_total=npairs, pairs=self.pairs[:min(npairs, self.capacity)].cpu().numpy().view(np.uint32),
                    static_pair_total=nst,
                    static_pairs=self.static_pairs[:min(nst, self.capacity)].cpu().numpy().view(np.uint32))

    def integrate_algorithmic_bytes(self):
        return 232 * self.n
