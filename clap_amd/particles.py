"""Host-side mirror of the reference's particle-system interface over device arrays.

Names follow core/particle.h: ``particles_update`` (particle.c:89, the per-entity update hook,
here batched over all systems), ``particle_system_count``, ``particle_system_position``,
and ``pos_array`` (what ``particle_system_upload`` hands to the shader, particle.c:122-125).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import synth


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class ParticleBatch:
    """All particle systems of a scene, device resident (include/clapgpu.h, clapgpu_particles)."""

    def __init__(self, ps, pos, vel, rng_state=synth.DRAND48_DEFAULT_STATE, device="cuda:0"):
        self.device = dev = torch.device(device)
        self.n = n = int(ps["n"])
        self.n_real = int(ps["n_real"])
        self.sys_host = np.ascontiguousarray(ps["sys"]).copy()
        self.n_sys = self.sys_host.shape[0]
        self.sys = torch.from_numpy(self.sys_host.view(np.uint8).reshape(self.n_sys, 64)).to(dev)
        self.row_sys = torch.from_numpy(np.ascontiguousarray(ps["row_sys"]).view(np.int32)).to(dev)
        self.pos = torch.from_numpy(np.ascontiguousarray(pos, np.float32)).to(dev)
        self.vel = torch.from_numpy(np.ascontiguousarray(vel, np.float32)).to(dev)
        self.rng_state = torch.tensor([rng_state, rng_state], dtype=torch.int64, device=dev)
        self.billboard_mx = torch.zeros((self.n_sys, 16), dtype=torch.float32, device=dev)
        rows = n // 64
        self.respawn_mask = torch.zeros((max(rows, 1),), dtype=torch.int64, device=dev)
        self.respawn_row_pop = torch.zeros(((rows + 15) // 16 * 16 or 16,), dtype=torch.uint8, device=dev)
        self.respawn_list = torch.zeros((max(n, 1),), dtype=torch.int32, device=dev)
        self.respawn_count = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.scratch = torch.zeros((_lib.lib().clapgpu_visible_scratch_bytes(n) // 4 or 1,), dtype=torch.int32,
                                   device=dev)
        self.respawn_groups = torch.zeros((132,), dtype=torch.int32, device=dev)    # CLAPGPU_RESPAWN_GROUP_WORDS
        self._desc = _lib.Particles(
            n=n, n_sys=self.n_sys, sys=self.sys.data_ptr(), row_sys=self.row_sys.data_ptr(),
            pos=self.pos.data_ptr(), vel=self.vel.data_ptr(), rng_state=self.rng_state.data_ptr(),
            billboard_mx=self.billboard_mx.data_ptr(), respawn_mask=self.respawn_mask.data_ptr(),
            respawn_row_pop=self.respawn_row_pop.data_ptr(), respawn_list=self.respawn_list.data_ptr(),
            respawn_count=self.respawn_count.data_ptr(), scratch=self.scratch.data_ptr(),
            respawn_groups=self.respawn_groups.data_ptr())

    def particles_update(self, view_mx):
        """particles_update for every system (mq order), one libc-compatible drand48 stream."""
        v = np.ascontiguousarray(view_mx, np.float32)
        rc = _lib.lib().clapgpu_particles_update(C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                 C.byref(self._desc), v.ctypes.data_as(C.POINTER(C.c_float)))
        _lib.check(rc, "clapgpu_particles_update")

    def particle_system_count(self, s):
        return int(self.sys_host["count"][s])

    def pos_array(self, s):
        """Device view of system s's pos_array (count x vec3), the UNIFORM_PARTICLE_POS payload."""
        f, c = int(self.sys_host["first"][s]), int(self.sys_host["count"][s])
        return self.pos[f:f + c]

    def particle_system_position(self, s, center, attached=False):
        """particle_system_position (particle.c:132-157): move the emitter; an attached system
        carries its particles along."""
        center = np.asarray(center, np.float32)
        old = self.sys_host["center"][s].copy()
        delta = center - old
        if attached and float(np.dot(delta, delta)) != 0.0:
            self.pos_array(s).add_(torch.from_numpy(delta).to(self.device))
        self.sys_host["center"][s] = center
        self.sys[s].copy_(torch.from_numpy(self.sys_host[s:s + 1].view(np.uint8).reshape(64)))

    def stream_state(self):
        """drand48 state after the last update (host sync)."""
        return int(self.rng_state[1].item()) & ((1 << 48) - 1)

    def download(self):
        torch.cuda.synchronize(self.device)
        return dict(pos=self.pos.cpu().numpy(), vel=self.vel.cpu().numpy(),
                    billboard_mx=self.billboard_mx.cpu().numpy(), rng_state=self.stream_state(),
                    respawned=int(self.respawn_count.item()))

    def algorithmic_bytes(self):
        return 36 * self.n_real      # pos 12 + vel 12 read, pos 12 written
