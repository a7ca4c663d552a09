"""SoA scene snapshot format (include/clapgpu_snapshot.h).  CPU: the C
writer / reader round trip, layout guarantees and rejection of damaged files.  GPU: a scene saved to
a snapshot and replayed through the kernels gives the oracle's results."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from clap_amd import _lib, snapshot, synth
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_functions_all_bound_and_exported():
    text = open(os.path.join(ROOT, "include", "clapgpu_snapshot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(clapgpu_\w+)\s*\(", text)))
    assert sorted(snapshot.SYMBOLS) == declared
    out = subprocess.run(["nm", "-D", "--defined-only", snapshot.SCENE_LIB_PATH], capture_output=True, text=True, check=True)
    exported = {l.split()[-1] for l in out.stdout.splitlines() if " T " in l}
    assert set(declared) <= exported


def test_round_trip_all_dtypes_and_shapes(tmp_path):
    rng = np.random.Generator(np.random.PCG64(1))
    arrays = {"u8": rng.integers(0, 255, (7, 3), dtype=np.uint8), "i32": rng.integers(-9, 9, 11).astype(np.int32),
              "u32.flags": rng.integers(0, 2**32, (5, 2, 2), dtype=np.uint32),
              "f32": rng.normal(size=(4, 3, 2, 5)).astype(np.float32), "f64": rng.normal(size=13),
              "u64": rng.integers(0, 2**63, 3, dtype=np.uint64), "i64": np.asarray([-5], np.int64),
              "empty": np.zeros((0, 3), np.float32), "scalar0d": np.asarray(2.5, np.float32)}
    p = str(tmp_path / "a.clps")
    snapshot.save(p, arrays)
    back = snapshot.load(p)
    assert list(back) == list(arrays), "order is kept"
    for k, a in arrays.items():
        assert back[k].dtype == a.dtype and back[k].shape == a.shape and np.array_equal(back[k], a), k
    raw = open(p, "rb").read()
    magic, version, n, table_off, size = struct.unpack_from("<8sIIQQ", raw)
    assert magic == b"CLAPSNP1" and version == 1 and n == len(arrays) and size == len(raw) and table_off % 64 == 0
    for k in range(n):
        off = struct.unpack_from("<Q", raw, table_off + 96 * k + 88)[0]
        assert off % 64 == 0, "every payload starts on a 64-byte boundary"


def test_writer_and_reader_reject_bad_input(tmp_path):
    L = snapshot.lib()
    with pytest.raises(ValueError):
        snapshot.save(str(tmp_path / "b.clps"), {"c": np.zeros(3, np.complex64)})
    assert not os.path.exists(tmp_path / "b.clps"), "an aborted write leaves no file"
    w = C.c_void_p()
    assert L.clapgpu_snapshot_create(C.byref(w), str(tmp_path / "d.clps").encode()) == 0
    dims = (C.c_uint64 * 4)(2, 0, 0, 0)
    buf = (C.c_float * 2)(1, 2)
    assert L.clapgpu_snapshot_add(w, b"x", 4, 1, dims, buf) == 0
    assert L.clapgpu_snapshot_add(w, b"x", 4, 1, dims, buf) == _lib.ERR_INVALID_ARGUMENTS, "duplicate name"
    assert L.clapgpu_snapshot_add(w, b"y" * 48, 4, 1, dims, buf) == _lib.ERR_INVALID_ARGUMENTS, "name too long"
    assert L.clapgpu_snapshot_add(w, b"z", 99, 1, dims, buf) == _lib.ERR_INVALID_ARGUMENTS, "unknown dtype"
    assert L.clapgpu_snapshot_add(w, b"r", 4, 5, dims, buf) == _lib.ERR_INVALID_ARGUMENTS, "rank > 4"
    assert L.clapgpu_snapshot_finish(w) == 0
    good = open(tmp_path / "d.clps", "rb").read()

    def opens(blob):
        q = tmp_path / "t.clps"
        q.write_bytes(blob)
        s = C.c_void_p()
        rc = L.clapgpu_snapshot_open(C.byref(s), str(q).encode())
        if rc == 0:
            L.clapgpu_snapshot_close(s)
        return rc
    assert opens(good) == 0
    assert opens(b"NOTASNAP" + good[8:]) != 0, "magic"
    assert opens(good[:12] + struct.pack("<I", 7) + good[16:]) != 0, "array count does not match the table"
    assert opens(good[:-8]) != 0, "truncated"
    assert opens(good + b"\0" * 8) != 0, "trailing bytes"
    table_off = struct.unpack_from("<Q", good, 16)[0]
    huge = bytearray(good)
    struct.pack_into("<Q", huge, table_off + 56, 1 << 40)           # dims[0] of the first array
    assert opens(bytes(huge)) != 0, "array larger than the file"
    assert opens(b"") != 0 and opens(good[:20]) != 0
    s = C.c_void_p()
    assert L.clapgpu_snapshot_open(C.byref(s), str(tmp_path / "missing.clps").encode()) != 0


def test_c_reader_writer_under_sanitizers(tmp_path):
    """tests/c/test_snapshot.c built with AddressSanitizer + UBSan (host code only): round trip from C,
    and bit flips over the whole header and table must be refused or stay in bounds."""
    exe = str(tmp_path / "test_snapshot_c")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_snapshot.c"),
                    os.path.join(ROOT, "clap_amd", "host", "clapgpu_snapshot.c"), "-o", exe], check=True)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout


def test_scene_components_round_trip(tmp_path):
    scene = synth.pad_levels(synth.entities_forest(300, seed=4))
    cam = synth.camera(pos=(1, 2, 3))
    bodies = synth.sphere_bodies(50, box=8.0, seed=2)
    p = str(tmp_path / "scene.clps")
    snapshot.save_scene(p, entities=scene, camera=cam, bodies=bodies, lights=synth.lights(8, seed=1))
    back = snapshot.load_scene(p)
    assert set(back) == {"entities", "camera", "bodies", "lights"}
    assert back["entities"]["n"] == scene["n"] and back["bodies"]["cell"] == bodies["cell"]
    for k, v in scene.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(back["entities"][k], v) and back["entities"][k].dtype == v.dtype, k


@pytest.mark.gpu
def test_replay_from_snapshot_matches_oracle(tmp_path, cuda_device):
    """What a CLAP-side dump is for: entities + camera written once, loaded, run on the GPU."""
    from clap_amd import entities, tiler
    base = synth.entities_forest(20_000, seed=8, max_depth=6)
    scene, _perm = tiler.tiled_scene(base)
    cam = synth.camera(pos=(0, 5, 40))
    p = str(tmp_path / "replay.clps")
    snapshot.save_scene(p, entities=scene, camera=cam)
    back = snapshot.load_scene(p)
    batch = entities.EntityBatch(back["entities"], cuda_device)
    fr, _v, _p = entities.view_calc_frustum(back["camera"])
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    out = batch.download()
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    ofr, _ov, _op = ob.frustum_from_camera(cam)
    vis, _m = ob.entities_cull(scene["n"], st["flags"], st["aabb"], ofr)
    assert np.array_equal(out["mx"].view(np.uint32), st["mx"].view(np.uint32))
    assert np.array_equal(out["visible"], vis)
