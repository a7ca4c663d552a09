"""Host-side mirror of the reference's physics interface for capsule / sphere bodies without constraint rows.

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
    def __init__(self, bodies, statics=None, pair_capacity=None, device="cuda:0", static_pair_capacity=None, geom_records=True):
        """bodies: dict from synth.sphere_bodies() / synth.capsule_bodies(); statics: float64 [ns, 6]
        (minx,maxx,miny,maxy,minz,maxz), host array: binned once by clapgpu_bp_create."""
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
        self.length = t("length", np.float64) if "length" in bodies else None
        self.inertia = t("inertia", np.float64) if "inertia" in bodies else None
        self.aabb = torch.zeros((max(n, 1), 6), dtype=torch.float64, device=dev)
        self.axis = torch.zeros((max(n, 1), 3), dtype=torch.float64, device=dev)
        self.cell = float(bodies["cell"])
        self.world = _lib.World()
        _lib.lib().clapgpu_world_defaults(C.byref(self.world))
        self.time_acc = C.c_double(0.0)
        samples = int(bodies.get("adis_average_samples", 1))
        self.adis_samples = self.adis_counter = None
        if samples > 1:
            self.adis_samples = torch.zeros((max(n, 1), samples, 6), dtype=torch.float64, device=dev)
            self.adis_counter = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        d = _lib.Bodies(n, samples, _ptr(self.pos), _ptr(self.quat), _ptr(self.lvel), _ptr(self.avel),
                        _ptr(self.mass), _ptr(self.radius), _ptr(self.yoffset), _ptr(self.bflags),
                        _ptr(self.adis_steps_left), _ptr(self.adis_time_left), _ptr(self.body_entity))
        d.length, d.inertia = _ptr(self.length), _ptr(self.inertia)
        _lib.lib().clapgpu_geom_offset_rotation(d.geom_offset_R)           # physics.c:974-978
        d.aabb, d.axis = _ptr(self.aabb), _ptr(self.axis)
        d.adis_samples, d.adis_counter = _ptr(self.adis_samples), _ptr(self.adis_counter)
        # the narrowphase's one-sector view of every body geom (clapgpu_bodies.geom_records), kept by the step / aabb kernels
        self.geom_records = torch.zeros((max(n, 1), 8), dtype=torch.float64, device=dev) if geom_records else None
        d.geom_records = _ptr(self.geom_records)
        self._desc = d
        self.capacity = int(pair_capacity if pair_capacity is not None else max(8 * n, 1024))
        self.static_capacity = int(static_pair_capacity if static_pair_capacity is not None else self.capacity)
        self.pairs = torch.zeros((self.capacity, 2), dtype=torch.int32, device=dev)
        self.pair_total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.static_pairs = torch.zeros((self.static_capacity, 2), dtype=torch.int32, device=dev)
        self.static_pair_total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_static = 0
        st = None
        if statics is not None and len(statics):
            st = np.ascontiguousarray(statics, np.float64)
            self.n_static = st.shape[0]
        self._statics_host = st
        self._bp = C.c_void_p()
        _lib.check(_lib.lib().clapgpu_bp_create(C.byref(self._bp), n, self.cell, self.n_static,
                                                st.ctypes.data if st is not None else None), "clapgpu_bp_create")
        self.statics_ptr = _lib.lib().clapgpu_bp_static_aabb(self._bp)     # device copy owned by the broadphase object
        self.bodies_aabb()

    def __del__(self):
        bp, self._bp = getattr(self, "_bp", None), None
        if bp:
            try:
                _lib.lib().clapgpu_bp_destroy(bp)
            except Exception:
                pass

    # ---- __phys_step pieces -----------------------------------------------------------
    def bp_invalidate(self):
        """Boxes a step pre-binned (world_step(prebin=True) / FrameLoop(prebin=True)) were rewritten by something else."""
        _lib.check(_lib.lib().clapgpu_bp_invalidate(_stream(), self._bp), "clapgpu_bp_invalidate")

    def bodies_aabb(self):
        self.bp_invalidate()
        """Geom axis + AABB of every body from its pose (after the host moved bodies; world_step keeps them current)."""
        _lib.check(_lib.lib().clapgpu_bodies_aabb(_stream(), C.byref(self._desc)), "clapgpu_bodies_aabb")

    def broadphase(self, side=None):
        """dSpaceCollide2(ground, bodies) + dSpaceCollide(bodies) (physics.c:751-753): both candidate pair lists,
        ascending, from one pass of four launches over the bodies' AABBs."""
        L = _lib.lib()
        _lib.check(L.clapgpu_bp_collide(_stream(), self._bp, self.n, _ptr(self.aabb), _ptr(self.pairs), self.capacity,
                                        _ptr(self.pair_total), _ptr(self.static_pairs) if self.n_static else None,
                                        self.static_capacity if self.n_static else 0,
                                        _ptr(self.static_pair_total) if self.n_static else None), "clapgpu_bp_collide")

    def broadphase_status(self):
        st = C.c_uint32(0)
        _lib.check(_lib.lib().clapgpu_bp_status(_stream(), self._bp, C.byref(st)), "clapgpu_bp_status")
        return st.value

    def body_geoms(self):
        g = _lib.Geoms(self.n, 0, _ptr(self.pos), _ptr(self.axis), _ptr(self.radius), _ptr(self.length), 0, 0,
                       _ptr(getattr(self, "material", None)), _ptr(self.geom_records))
        return g

    def static_geoms(self):
        """The statics as axis-aligned boxes (their AABBs); static_geom_arrays overrides kind / pos / radius / ..."""
        sg = getattr(self, "_static_geoms", None)
        if sg is None:
            kind = torch.full((max(self.n_static, 1),), _lib.GEOM_BOX, dtype=torch.uint8, device=self.device)
            self._static_keep = dict(kind=kind)
            sg = _lib.Geoms(self.n_static, 0, 0, 0, 0, 0, _ptr(kind), self.statics_ptr,
                            _ptr(getattr(self, "static_material", None)))
            self._static_geoms = sg
        sg.material = _ptr(getattr(self, "static_material", None))
        return sg

    def set_static_geoms(self, kind, pos=None, axis=None, radius=None, length=None):
        """Narrowphase description of the statics (default: every static is its AABB as a box)."""
        dev = self.device
        keep = dict(kind=torch.from_numpy(np.ascontiguousarray(kind, np.uint8)).to(dev))
        for name, a in (("pos", pos), ("axis", axis), ("radius", radius), ("length", length)):
            keep[name] = None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev)
        self._static_keep = keep
        self._static_geoms = _lib.Geoms(self.n_static, 0, _ptr(keep["pos"]), _ptr(keep["axis"]), _ptr(keep["radius"]),
                                        _ptr(keep["length"]), _ptr(keep["kind"]), self.statics_ptr,
                                        _ptr(getattr(self, "static_material", None)))

    def alloc_contacts(self):
        if getattr(self, "contact2_buf", None) is None:
            self.contact2_buf = torch.zeros((self.capacity, 160), dtype=torch.uint8, device=self.device)
            self.contact2_total = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.static_contact2_buf = torch.zeros((self.static_capacity if self.n_static else 1, 160), dtype=torch.uint8,
                                                   device=self.device)
            self.static_contact2_total = torch.zeros(1, dtype=torch.int32, device=self.device)

    def upload_links(self, link_body, link_entity):
        """Device copies of a (body, entity) link table, uploaded once per table."""
        key = (id(link_body), id(link_entity))
        if getattr(self, "_links_key", None) != key:
            self._links_key = key
            self._links = (torch.from_numpy(np.ascontiguousarray(link_body, np.uint32).view(np.int32)).to(self.device),
                           torch.from_numpy(np.ascontiguousarray(link_entity, np.uint32).view(np.int32)).to(self.device))
        return self._links

    def contacts_geoms(self, set_joint_flags=True):
        """near_callback on both candidate lists of the last broadphase(): 160-byte records (clapgpu_contact2)."""
        L = _lib.lib()
        self.alloc_contacts()
        g = self.body_geoms()
        fl = _ptr(self.bflags) if set_joint_flags else 0
        _lib.check(L.clapgpu_contacts_geoms(_stream(), C.byref(g), C.byref(g), _ptr(self.pairs), _ptr(self.pair_total),
                                            self.capacity, _ptr(self.contact2_buf), _ptr(self.contact2_total), fl, fl),
                   "clapgpu_contacts_geoms")
        if self.n_static:
            sg = self.static_geoms()
            _lib.check(L.clapgpu_contacts_geoms(_stream(), C.byref(g), C.byref(sg), _ptr(self.static_pairs),
                                                _ptr(self.static_pair_total), self.static_capacity,
                                                _ptr(self.static_contact2_buf), _ptr(self.static_contact2_total), fl, 0),
                       "clapgpu_contacts_geoms(static)")

    def contacts_geoms_both(self, set_joint_flags=True):
        """contacts_geoms() as ONE launch over both lists (clapgpu_contacts_geoms_both): needs statics."""
        L = _lib.lib()
        self.alloc_contacts()
        g, sg = self.body_geoms(), self.static_geoms()
        fl = _ptr(self.bflags) if set_joint_flags else 0
        _lib.check(L.clapgpu_contacts_geoms_both(_stream(), self._bp, C.byref(g), C.byref(sg), _ptr(self.pairs),
                                                 _ptr(self.pair_total), self.capacity, _ptr(self.contact2_buf),
                                                 _ptr(self.contact2_total), _ptr(self.static_pairs),
                                                 _ptr(self.static_pair_total), self.static_capacity,
                                                 _ptr(self.static_contact2_buf), _ptr(self.static_contact2_total), fl),
                   "clapgpu_contacts_geoms_both")

    def download_contacts2(self, dtype):
        torch.cuda.synchronize(self.device)
        npairs = min(int(self.pair_total.item()), self.capacity)
        out = dict(body=(self.contact2_buf[:npairs].cpu().numpy().view(dtype).reshape(-1), int(self.contact2_total.item())))
        if self.n_static:
            ns = min(int(self.static_pair_total.item()), self.static_capacity)
            out["static"] = (self.static_contact2_buf[:ns].cpu().numpy().view(dtype).reshape(-1),
                             int(self.static_contact2_total.item()))
        return out

    def sweep_capsules(self, sweep_body, delta, cand_first, cand):
        """phys_body_sweep_capsule for a batch (physics.c:559-670): returns (frac, normal[n,3], hit) device tensors."""
        dev = self.device
        ns = len(sweep_body)
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt).view(np.int32 if dt == np.uint32 else dt)).to(dev)
        sb, dl, cf, cd = up(sweep_body, np.uint32), up(delta, np.float32), up(cand_first, np.uint32), up(cand, np.uint32)
        frac = torch.zeros(max(ns, 1), dtype=torch.float32, device=dev)
        normal = torch.zeros((max(ns, 1), 3), dtype=torch.float32, device=dev)
        hit = torch.zeros(max(ns, 1), dtype=torch.int32, device=dev)
        g, sg = self.body_geoms(), self.static_geoms()
        _lib.check(_lib.lib().clapgpu_sweep_capsules(_stream(), C.byref(g), C.byref(sg), ns, _ptr(sb), _ptr(dl), _ptr(cf),
                                                     _ptr(cd), _ptr(frac), _ptr(normal), _ptr(hit)), "clapgpu_sweep_capsules")
        return frac[:ns], normal[:ns], hit[:ns]

    def rotate_from_entities(self, entity_batch, link_body, link_entity, all_dirty=False):
        """phys_body_rotate_xform for the (body, entity) links whose entity default_update is about to
        rebuild (model.c:1680-1687); run before entity_batch.mq_update."""
        lb, le = self.upload_links(link_body, link_entity)     # once per link table (also keeps graph capture clean)
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
            self.static_contact_buf = torch.zeros((self.static_capacity, 104), dtype=torch.uint8, device=self.device)
            self.static_contact_total = torch.zeros(1, dtype=torch.int32, device=self.device)
        if static_material is not None:
            self.static_material = torch.from_numpy(np.ascontiguousarray(static_material, np.float64)).to(self.device)
        mat, smat = getattr(self, "material", None), getattr(self, "static_material", None)
        _lib.check(_lib.lib().clapgpu_contacts_sphere_box(_stream(), C.byref(self._desc), self.n_static,
                                                          self.statics_ptr, _ptr(self.static_pairs),
                                                          _ptr(self.static_pair_total), self.static_capacity, _ptr(mat),
                                                          _ptr(smat), _ptr(self.static_contact_buf),
                                                          _ptr(self.static_contact_total)),
                   "clapgpu_contacts_sphere_box")

    def download_static_contacts(self, dtype):
        torch.cuda.synchronize(self.device)
        npairs = min(int(self.static_pair_total.item()), self.static_capacity)
        return (self.static_contact_buf[:npairs].cpu().numpy().view(dtype).reshape(-1),
                int(self.static_contact_total.item()))

    def download_contacts(self, dtype):
        torch.cuda.synchronize(self.device)
        npairs = min(int(self.pair_total.item()), self.capacity)
        return self.contact_buf[:npairs].cpu().numpy().view(dtype).reshape(-1), int(self.contact_total.item())

    def world_step(self, h, prebin=False):
        """quickstep's body stage; prebin: also the bin pass of the next broadphase() over the boxes it writes."""
        if prebin:
            _lib.check(_lib.lib().clapgpu_bodies_step_prebin(_stream(), C.byref(self._desc), C.byref(self.world), h, self._bp),
                       "clapgpu_bodies_step_prebin")
            return
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
                    avel=self.avel.cpu().numpy(), aabb=self.aabb.cpu().numpy()[:self.n], axis=self.axis.cpu().numpy()[:self.n],
                    bflags=self.bflags.cpu().numpy().view(np.uint32),
                    adis_steps_left=self.adis_steps_left.cpu().numpy(), adis_time_left=self.adis_time_left.cpu().numpy(),
                    pair_total=npairs, pairs=self.pairs[:min(npairs, self.capacity)].cpu().numpy().view(np.uint32),
                    static_pair_total=nst,
                    static_pairs=self.static_pairs[:min(nst, self.static_capacity)].cpu().numpy().view(np.uint32))

    def integrate_algorithmic_bytes(self):
        return 232 * self.n
