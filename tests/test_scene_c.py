"""The C host mirror (include/clapgpu_scene.h, clap_amd/host/clapgpu_scene.c).

CPU: it builds as plain C11, exports what its header declares, and a C program using it links.
GPU: tests/c/test_scene.c drives it like CLAP's frame loop (create in any order, move,
re-parent, delete, add, cull) and compares every frame with the oracle bit for bit."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "clap_amd", "lib")
SCENE_SO = os.path.join(LIBDIR, "libclapgpu_scene.so")
TEST_BIN = os.path.join(ROOT, "tests", "c", "_build", "test_scene")
ABI_BIN = os.path.join(ROOT, "tests", "c", "_build", "test_abi")


def build_c_test():
    from clap_amd import _lib
    from oracle import binding
    if not os.path.exists(SCENE_SO):
        _lib.build()
    binding.lib()
    os.makedirs(os.path.dirname(TEST_BIN), exist_ok=True)
    odir = os.path.join(ROOT, "oracle", "_build")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "c", "test_scene.c"), "-o", TEST_BIN,
                    "-L", LIBDIR, "-lclapgpu_scene", "-lclapgpu", "-L", odir, "-lclap_oracle", "-lm",
                    f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{odir}"], check=True)


def build_abi_test():
    from clap_amd import _lib
    from oracle import binding
    _lib.lib()
    binding.lib()
    os.makedirs(os.path.dirname(ABI_BIN), exist_ok=True)
    odir = os.path.join(ROOT, "oracle", "_build")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "c", "test_abi.c"), "-o", ABI_BIN,
                    "-L", LIBDIR, "-lclapgpu", "-L", odir, "-lclap_oracle", "-lm",
                    f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{odir}"], check=True)


def test_scene_library_exports_its_header():
    from clap_amd import _lib
    if not os.path.exists(SCENE_SO):
        _lib.build()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "clapgpu_scene.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(clapgpu_scene_[a-z0-9_]+)\s*\(", src)))
    assert len(declared) >= 18
    out = subprocess.run(["nm", "-D", "--defined-only", SCENE_SO], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared) <= exported, sorted(set(declared) - exported)


def test_c_program_links_against_the_host_mirror():
    build_c_test()
    assert os.access(TEST_BIN, os.X_OK)


def test_c_caller_of_the_flat_abi_builds():
    build_abi_test()
    assert os.access(ABI_BIN, os.X_OK)


@pytest.mark.gpu
def test_c_caller_of_the_flat_abi_matches_oracle(cuda_device):
    """particles, bodies + broadphase + contacts + read-back, light grid: driven from plain C."""
    build_abi_test()                                        # always from the current sources (a second's work)
    r = subprocess.run([ABI_BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout and all(k in r.stdout for k in ("particles ok", "bodies ok", "light grid ok"))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["small_frames", "staged"])
@pytest.mark.parametrize("mode", ["tiles", "wide"])
def test_scene_mirror_frames_match_oracle(mode, path, cuda_device):
    """Five frames of creations / moves / re-parenting / deletions, four of creations / deletions that edit the standing
    layout in place (clapgpu_scene_entity_new_placed / _delete_placed), then three with joint riders through the frame's second
    launch (clapgpu_scene_attached_update), every entity bit for bit against the oracle -- on the small-frame path (touched
    records in through mapped memory, rebuilt rows out, polled completion word) and staged through device slabs."""
    build_c_test()
    env = dict(os.environ, CLAPGPU_SCENE_ZERO_COPY_SLOTS="0" if path == "staged" else "4294967295")
    r = subprocess.run([TEST_BIN] + (["wide"] if mode == "wide" else []), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout and r.stdout.count("frame ok") == 12
    # the four frames of in-place creations / deletions: where the layout takes them (one-launch tile form) most fit
    assert "layout edited in place" in r.stdout
    if mode != "wide" and path == "small_frames":
        assert " 0 entities placed" not in r.stdout
    assert ("small-frame path" if path == "small_frames" else "staged path") in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8])
def test_scene_mirror_long_run_of_in_place_edits(seed, cuda_device):
    """120 frames of creations and deletions that edit the standing layout in place (growth tiles, recycled lanes, rows that
    fill up and fall back to a re-tile, then in place again), every frame and every LOD pass against the oracle."""
    build_c_test()
    r = subprocess.run([TEST_BIN, "tiles", str(seed), "120"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, CLAPGPU_SCENE_ZERO_COPY_SLOTS="4294967295"))
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("frame ok") == 8 + 120
    assert " 0 entities placed" not in r.stdout and "of 120 frames without one" in r.stdout


def test_quat_from_angles_matches_reference(golden_dir):
    """transform_set_angles (clamp, degrees, euler xyz -> quaternion): the host helper against the
    reference's own function, bit-exact (same libm)."""
    import ctypes as C
    import numpy as np
    from clap_amd import snapshot
    L = snapshot.lib()                                      # loads libclapgpu_scene.so
    L.clapgpu_quat_from_angles.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    L.clapgpu_quat_from_angles.restype = None
    z = np.load(os.path.join(golden_dir, "transform_verbs.npz"))
    fp = C.POINTER(C.c_float)
    q = np.zeros(4, np.float32)
    for i in range(len(z["in_angles"])):
        a = np.ascontiguousarray(z["in_angles"][i])
        L.clapgpu_quat_from_angles(a.ctypes.data_as(fp), int(z["in_degrees"][i]), q.ctypes.data_as(fp))
        assert np.array_equal(q.view(np.uint32), z["ref_quat"][i].view(np.uint32)), (i, a, q, z["ref_quat"][i])
    # transform_move is a plain fp32 add
    assert np.array_equal((z["in_pos"] + z["in_off"]).view(np.uint32), z["ref_pos"].view(np.uint32))
