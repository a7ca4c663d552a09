"""Broadphase A/B helper: the pair SETS (and the order-insensitive hash of the lists) of a few scenes, written to
gpurun_out/bp_sets_<tag>.npz; run once with the shipped library, once with an experiment build
(CLAPGPU_LIB=clap_amd/lib_exp/libclapgpu.so CLAPGPU_ALLOW_EXPERIMENT=1) and compare with `cmp` mode."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
        print(k, a[k].shape, b[k].shape, "equal" if same else "DIFFERENT")
        ok &= same
    sys.exit(0 if ok else 1)
import torch
from clap_amd import _lib, physics, synth
_lib.check(_lib.lib().clapgpu_init(0), "init")
out = {}
for kind, n in (("spheres", 262_144), ("capsules", 262_144), ("spheres", 5000), ("pile", 3000)):
    if kind == "pile":
        b = synth.sphere_bodies(n, box=64.0, seed=9)
        b["pos"][:1500] = b["pos"][0] + np.random.default_rng(1).uniform(-0.3, 0.3, (1500, 3))
    elif kind == "spheres":
        b = synth.sphere_bodies(n, box=64.0 if n > 10000 else 12.0, seed=4)
    else:
        b = synth.capsule_bodies(n, box=60.0, seed=4)
    pw = physics.PhysWorld(b, synth.static_boxes(64, 64.0), pair_capacity=4_000_000, device="cuda:0")
    for rep in range(3):
        pw.broadphase(); torch.cuda.synchronize()
    nb, ns = int(pw.pair_total.item()), int(pw.static_pair_total.item())
    pb = pw.pairs[:nb].cpu().numpy().astype(np.int64); ps = pw.static_pairs[:ns].cpu().numpy().astype(np.int64)
    ub, us = np.unique(pb[:, 0] << 32 | pb[:, 1]), np.unique(ps[:, 0] << 32 | ps[:, 1])
    print(kind, n, "pairs", nb, "unique", len(ub), "static", ns, "unique", len(us), "status", pw.broadphase_status())
    assert nb == len(ub) and ns == len(us), "duplicates in the list"
    out[f"{kind}_{n}_body"] = ub; out[f"{kind}_{n}_static"] = us
np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"bp_sets_{sys.argv[1]}.npz"), **out)
