"""Host-side mirror of the reference's per-character update hook.

``CharacterFeed.character_update`` is ``character_update`` (character.c:583-611) for every character
of the scene, minus its tail call: the limbo teleport out of the position history, the body
read-back and ``history_push``.  The tail call (``orig_update`` = ``default_update``) is
``EntityBatch.mq_update``; ``character_motion_reset`` / ``character_move`` act on the controlled
character or call ODE sweeps and stay with the host.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class CharacterFeed:
    def __init__(self, feed, device="cuda:0"):
        """feed: dict as made by clap_amd.synth.character_feed() (entity, body, hist_*, airborne, limbo_height)."""
        self.device = dev = torch.device(device)
        self.n = n = int(feed["n"])
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
        self.entity = t(np.asarray(feed["entity"], np.uint32).view(np.int32), np.int32)
        self.body = t(feed["body"], np.int32)
        self.hist_pos = t(feed["hist_pos"], np.float32)
        self.hist_head = t(np.asarray(feed["hist_head"], np.uint32).view(np.int32), np.int32)
        self.hist_wrapped = t(feed["hist_wrapped"], np.uint8)
        self.airborne = t(feed["airborne"], np.uint8)
        self.moved = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        self.limbo_height = float(feed["limbo_height"])
        self._desc = _lib.Characters(n, self.limbo_height, self.entity.data_ptr(), self.body.data_ptr(),
                                     self.hist_pos.data_ptr(), self.hist_head.data_ptr(),
                                     self.hist_wrapped.data_ptr(), self.airborne.data_ptr(), self.moved.data_ptr())

    def set_airborne(self, airborne):
        """character.airborne as the host's character_move left it."""
        self.airborne.copy_(torch.from_numpy(np.ascontiguousarray(airborne, np.uint8)))

    def character_update(self, batch, world=None):
        """batch: EntityBatch; world: PhysWorld holding the characters' bodies, or None."""
        rc = _lib.lib().clapgpu_characters_update(_stream(), C.byref(self._desc), C.byref(batch._desc),
                                                  C.byref(world._desc) if world is not None else None)
        _lib.check(rc, "clapgpu_characters_update")

    def download(self):
        torch.cuda.synchronize(self.device)
        return dict(hist_pos=self.hist_pos.cpu().numpy(), hist_head=self.hist_head.cpu().numpy().view(np.uint32),
                    hist_wrapped=self.hist_wrapped.cpu().numpy(), moved=self.moved.cpu().numpy()[:self.n])
