import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from clap_amd import _lib, physics, synth
_lib.check(_lib.lib().clapgpu_init(0),"init")
for kind in (sys.argv[1:] or ("spheres","capsules")):
    b = synth.sphere_bodies(262_144, box=64.0, seed=4) if kind=="spheres" else synth.capsule_bodies(262_144, box=60.0, seed=4)
    pw = physics.PhysWorld(b, synth.static_boxes(64, 64.0 if kind=="spheres" else 60.0), pair_capacity=2_000_000, device="cuda:0")
    def t(fn, it=30):
        for _ in range(5): fn()
        ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
        torch.cuda.synchronize()
        for a,c in ev: a.record(); fn(); c.record()
        torch.cuda.synchronize()
        return np.mean([a.elapsed_time(c) for a,c in ev])*1e3
    print(kind, "broadphase us", t(pw.broadphase), "pairs", int(pw.pair_total.item()), "static pairs", int(pw.static_pair_total.item()),
          "contacts us", t(pw.contacts_geoms), "contacts (both lists, one launch) us", t(pw.contacts_geoms_both),
          "step us", t(lambda: pw.world_step(1/120)))
