"""One display frame of the batched path, in the reference's order.

``clap_frame()`` (core/clap.c:551-665) runs: ``phys_step`` (clap.c:604) -> ``scene_update`` ->
``mq_update`` with every entity's update hook in list order -- ``character_update`` (character.c:583)
in front of ``default_update`` (model.c:1649: body read-back, TRS rebuild, light hand-off, rotation
push to colliders, ``animated_update``), ``particles_update`` (particle.c:89) -> camera / light grid
-> render passes (frustum test, LOD pick per drawn entity; the vertex shader skins).  ``FrameLoop``
issues the same work as a fixed sequence of launches on one stream; nothing is read back unless asked.
``overlap=True`` puts the two pieces that depend on nothing else in the frame on a side stream next to
the physics chain (the statics x bodies broadphase pass beside the bodies x bodies pass, and the
particle systems), forking from and joining the main stream so the frame stays one unit and still captures
into one HIP graph.  Results are identical (tests/test_frame_gpu.py), but it is off by default: measured on
MI355X it does not shorten the BASELINE-size frame (0.604 vs 0.59 ms issued, 0.595 vs 0.60 ms replayed) and
the extra stream switches cost the testbed-size frame 80 us issued / 40 us replayed.
"""
import numpy as np
import torch

from . import entities as ent_mod


class FrameLoop:
    def __init__(self, batch, cam, world=None, feed=None, body_links=None, lights=None, characters=None,
                 particles=None, contacts=False, overlap=False):
        """batch: EntityBatch.  world: PhysWorld (dynamic bodies write their entities through
        body_entity; character bodies have body_entity = -1).  feed: CharacterFeed.  body_links:
        (link_body, link_entity) of characters / static colliders whose rotation follows the entity.
        lights: LightSet (with carriers).  characters: CharacterBatch (pose + skin).  particles:
        ParticleBatch."""
        self.batch, self.world, self.feed, self.lights = batch, world, feed, lights
        self.characters, self.particles = characters, particles
        self.body_links, self.contacts = body_links, contacts
        self.overlap, self._side = overlap, None
        self.set_camera(cam)

    def set_camera(self, cam):
        self.cam = cam
        self.frustum, self.view_mx, self.proj_mx = ent_mod.view_calc_frustum(cam)

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
        b, w = self.batch, self.world
        side = None
        if self.overlap:
            if self._side is None:
                self._side = torch.cuda.Stream()
            side = self._side
            main = torch.cuda.current_stream()
            side.wait_stream(main)                          # fork: everything of the previous frame is done
            if self.particles is not None:                  # particles_update hooks: independent of the rest of the frame
                with torch.cuda.stream(side):
                    self.particles.particles_update(self.view_mx)
        if w is not None:                                   # phys_step: per fixed substep broadphase, contacts, integrate
            for _ in range(steps):
                w.broadphase()
                if self.contacts:
                    w.contacts_geoms()
                w.world_step(1.0 / 120.0)
        if self.feed is not None:                           # character_update hooks
            self.feed.character_update(b, w)
        if w is not None:                                   # default_update: phys_body_update of dynamic bodies
            w.phys_body_update(b)
            if self.body_links is not None:                 # ... phys_body_rotate_xform for the rebuilt ones
                w.rotate_from_entities(b, *self.body_links)
        if self.lights is not None:                         # ... light_set_pos of light carriers
            self.lights.from_entities(b)
        b.mq_update(self.frustum)                           # TRS -> mx -> inverse -> AABB (+ main-view cull)
        if self.characters is not None:                     # ... animated_update: clock, pose, palette
            self.characters.animated_update(now)
            if self.characters._skin_desc is not None:
                self.characters.skin()                      # the vertex shader's skinning loop, once per frame
        if self.particles is not None and side is None:     # particles_update hooks
            self.particles.particles_update(self.view_mx)
        if self.lights is not None:                         # scene_update: light_grid_compute
            self.lights.grid_compute(self.view_mx, self.proj_mx)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)   # join
        b.compact_visible()                                 # render pass: visible list + LOD pick
        b.select_lod(self.cam["cam_pos"])
