"""Experiment builds with -DBP_EXP_TIMING: per-phase cycles of k_bp_search summed over its workgroups (thread 0 of each)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
from clap_amd import _lib, physics, synth
_lib.check(_lib.lib().clapgpu_init(0), "init")
names = ["prologue", "counts+select", "halo counts", "prefix", "stage", "rows (thread 0's wave)", "rows barrier wait", "tail"]
dbg = _lib.lib().clapgpu_bp_debug_ctrl
dbg.argtypes = [C.c_void_p, C.c_void_p]
for kind in (sys.argv[1:] or ("spheres", "capsules")):
    b = synth.sphere_bodies(262_144, box=64.0, seed=4) if kind == "spheres" else synth.capsule_bodies(262_144, box=60.0, seed=4)
    pw = physics.PhysWorld(b, synth.static_boxes(64, 64.0 if kind == "spheres" else 60.0), pair_capacity=2_000_000, device="cuda:0")
    for _ in range(3): pw.broadphase()
    buf = (C.c_uint32 * 160)()
    dbg(pw._bp, buf)
    a0 = np.array(buf[32:48], dtype=np.int64)
    reset = (C.c_uint32 * 4)(0xffffffff, 0, 0, 0)
    import torch as _t
    _lib.lib().clapgpu_bp_debug_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    _lib.lib().clapgpu_bp_debug_set(pw._bp, 48, reset, 4)
    pw.broadphase()
    dbg(pw._bp, buf)
    a1 = np.array(buf[32:48], dtype=np.int64)
    d = (a1 - a0) * 16
    print(kind, "ctrl", list(buf[:12]))
    print(f"   wall (100 MHz): first start .. last end {(buf[49]-buf[48])/100:.1f} us, longest workgroup {buf[50]/100:.1f} us, mean {buf[51]/512/100:.1f} us")
    for i, nm in enumerate(names):
        print(f"   {nm:28s} {d[i]/512:12.0f} cycles per workgroup (sum / 512)")
