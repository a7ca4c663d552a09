"""Where a one-launch small frame (clapgpu_entities_update_tiles_hostio) spends its time: 10 k entities, HIP events.
  none     nothing touched, nothing dirty      (row walk + masks + completion word)
  dirty    10 % DIRTY in the device flags      (+ rebuild + the rows' stores into mapped host memory)
  touched  10 % flagged in the mapped image    (+ the touched lanes' inputs read over PCIe)
  plain    clapgpu_entities_update_tiles, 10 % dirty (device memory only, no completion word)"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from clap_amd import _lib, entities, synth, tiler            # noqa: E402
from test_hostio_gpu import Hostio, Mapped, run_hostio       # noqa: E402

n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 7
L = _lib.lib()
scene = tiler.tiled_scene(synth.entities_forest(n_ent, 29, max_depth=depth))[0]
n, words = int(scene["n"]), int(scene["n"]) // 64
fr, _v, _p = entities.view_calc_frustum(synth.camera(pos=(0, 0, 80)))
b = entities.EntityBatch(scene, "cuda:0")
reb = torch.zeros(words + 2, dtype=torch.int64, device=b.device)
b._desc.rebuilt_mask = reb.data_ptr()
img, out, word = Mapped(n * 36 + (words + 2) * 8), Mapped(n * 164 + 3 * (words + 2) * 8), Mapped(64)
counter = torch.zeros(1, dtype=torch.int32, device=b.device)
h_ps, h_rot = img.view(0, 4 * n, np.float32).reshape(n, 4), img.view(16 * n, 4 * n, np.float32).reshape(n, 4)
h_fl, h_touched = img.view(32 * n, n, np.uint32), img.view(36 * n, words + 2, np.uint64)
h_ps[:], h_rot[:], h_fl[:] = scene["pos_scale"], scene["rot"], scene["flags"]
io = Hostio(pos_scale=img.dev(0), rot=img.dev(16 * n), flags=img.dev(32 * n), touched=0,
            mx=out.dev(0), inv_mx=out.dev(64 * n), aabb=out.dev(128 * n), center=out.dev(152 * n),
            vis_mask=out.dev(164 * n), rebuilt_mask=out.dev(164 * n + (words + 2) * 8), inside_mask=0,
            counter=counter.data_ptr(), done=word.dev(0))
rng = np.random.Generator(np.random.PCG64(1))
alive = (scene["flags"] & np.uint32(_lib.E_ALIVE)) != 0
sel = (rng.uniform(0, 1, n) < frac) & alive
idx = torch.as_tensor(np.flatnonzero(sel), device=b.device)
h_fl[sel] |= np.uint32(_lib.E_DIRTY)
tb = np.packbits(sel, bitorder="little").view(np.uint64)
h_touched[:len(tb)] = tb
fid = [0]


def frame(mode):
    if mode in ("dirty", "plain"):
        b.flags[idx] |= np.int32(_lib.E_DIRTY)
    if mode == "plain":
        b.mq_update(fr)
        return
    io.touched = img.dev(36 * n) if mode == "touched" else 0
    fid[0] += 1
    run_hostio(b, io, fr, fid[0])


def timed(mode, reps=200):
    for _ in range(20):
        frame(mode)
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, z in ev:
        if mode in ("dirty", "plain"):
            b.flags[idx] |= np.int32(_lib.E_DIRTY)
        a.record()
        if mode == "plain":
            b.mq_update(fr)
        else:
            io.touched = img.dev(36 * n) if mode == "touched" else 0
            fid[0] += 1
            run_hostio(b, io, fr, fid[0])
        z.record()
        torch.cuda.synchronize()
    t = sorted(a.elapsed_time(z) * 1e3 for a, z in ev)
    return t[len(t) // 2], t[len(t) // 10]


frame("dirty")
torch.cuda.synchronize()
print(f"{n_ent} entities ({n} slots, {b.n_tiles} tiles), {int(sel.sum())} selected")
for mode in ("none", "dirty", "touched", "plain"):
    med, p10 = timed(mode)
    print(f"  {mode:8s} median {med:6.1f} us   10th percentile {p10:6.1f} us")
