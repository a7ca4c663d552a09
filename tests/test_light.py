"""Clustered-lighting tile masks (light.c:88-154, 301-309).

CPU: the oracle against the reference's own light_grid_compute (golden fixtures captured from the
RGBA32UI buffer it uploads; live against oracle/_ref when present) and edge cases.
GPU: the HIP kernel through the C ABI against the oracle and the fixtures -- integer masks, bit-exact."""
import os

import numpy as np
import pytest

from clap_amd import synth
from oracle import binding as ob
from oracle import refrun

FIXTURES = ["lightgrid_1080p", "lightgrid_odd_z01"]
LIGHT_KEYS = ("pos", "color", "attenuation", "is_dir", "active")


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    lights = {k: z["in_" + k] for k in LIGHT_KEYS}
    lights["nr_lights"] = len(lights["active"])
    w, h, cell = (int(v) for v in z["in_grid"])
    return z, lights, w, h, cell


def _random_case(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    lights = synth.lights(int(rng.integers(1, 129)), seed=seed, extent=float(rng.uniform(5, 200)),
                          n_dir=int(rng.integers(0, 4)), inactive_frac=float(rng.uniform(0, 0.5)))
    cam = synth.camera(pos=rng.uniform(-20, 20, 3), quat=synth.quat_from_euler_xyz(*rng.uniform(-3, 3, 3)),
                       fov_deg=float(rng.uniform(30, 110)), ndc_z_zero_one=int(rng.integers(0, 2)))
    w, h = int(rng.integers(1, 2600)), int(rng.integers(1, 1500))
    cell = int(rng.choice([8, 16, 32, 64, 100]))
    return lights, cam, w, h, cell


# ---------------------------------------------------------------- CPU: oracle pinned on the reference
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_matches_reference_fixture(name, golden_dir):
    z, lights, w, h, cell = _load(golden_dir, name)
    tiles = ob.light_grid_compute(lights, z["ref_view_mx"], z["ref_proj_mx"], w, h, cell)
    assert tiles.shape == z["ref_tiles"].shape, "light_grid_update's tile counts"
    assert np.array_equal(tiles, z["ref_tiles"]), "RGBA32UI tile masks"
    rad = np.asarray([ob.light_radius(lights["color"][i], lights["attenuation"][i], lights["is_dir"][i])
                      for i in range(lights["nr_lights"])], np.float32)
    assert np.array_equal(rad.view(np.uint32), z["ref_radius"].view(np.uint32)), "light_get_radius"
    set_bits = np.unpackbits(tiles.view(np.uint8)).sum() / (tiles.shape[0] * tiles.shape[1])
    assert 2 < set_bits < 100, "the fixture exercises both outcomes of the disc test"


@pytest.mark.skipif(not refrun.available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_reference_live(seed):
    lights, cam, w, h, cell = _random_case(100 + seed)
    tiles, _rad, vm, pm = refrun.lightgrid(lights, cam, w, h, cell)
    assert np.array_equal(ob.light_grid_compute(lights, vm, pm, w, h, cell), tiles)


def test_oracle_edge_cases():
    _fr, vm, pm = ob.frustum_from_camera(synth.camera())
    assert ob.light_grid_dims(1920, 1080, 64) == (30, 17)
    assert ob.light_grid_dims(64, 64, 64) == (1, 1) and ob.light_grid_dims(65, 1, 64) == (2, 1)
    assert ob.light_grid_dims(100, 100, 0) == (0, 0)
    L = synth.lights(4, seed=1, n_dir=4, inactive_frac=0.0)
    t = ob.light_grid_compute(L, vm, pm, 640, 360, 64)
    assert np.all(t[..., 0] == 0xF) and not t[..., 1:].any(), "directional slots light every tile"
    L["active"][2] = 0
    t = ob.light_grid_compute(L, vm, pm, 640, 360, 64)
    assert np.all(t[..., 0] == 0xB), "a released slot contributes nothing"
    # a point light straight ahead lights the centre tile; far behind the far plane (ndc z > 1) or
    # on the eye plane (|w| < 1e-3) it is skipped
    P = dict(nr_lights=3, active=np.ones(3, np.uint32), is_dir=np.zeros(3, np.int32),
             pos=np.asarray([[0, 0, -10], [0, 0, -1e5], [0.3, 0.2, 0]], np.float32),
             color=np.ones((3, 3), np.float32), attenuation=np.tile(np.asarray([1, 2.0, 200.0], np.float32), (3, 1)))
    t = ob.light_grid_compute(P, vm, pm, 640, 384, 64)
    assert t[3, 5, 0] & 1 and not (t[..., 0] & 6).any()
    assert not (t[0, 0, 0] & 1), "its disc does not reach the screen corner"
    # slot 127 sets the top bit of the last word
    L = synth.lights(128, seed=2, n_dir=0, inactive_frac=0.0)
    L["is_dir"][127] = 1
    t = ob.light_grid_compute(L, vm, pm, 320, 200, 32)
    assert np.all(t[..., 3] >> 31 == 1)


def test_oracle_lights_from_entities():
    scene = synth.pad_levels(synth.entities_chains(40, 3, seed=5))
    L = synth.lights(16, seed=3, inactive_frac=0.0)
    L["active"][5] = 0
    n = scene["n"]
    roots = np.flatnonzero(scene["parent"] < 0)[:6]
    child = int(np.flatnonzero(scene["parent"] >= 0)[0])
    carriers = dict(entity=np.asarray([roots[0], roots[1], child, roots[2], roots[3], roots[4]], np.uint32),
                    light=np.asarray([3, 5, 6, 7, 3, 40], np.int32),
                    off=np.arange(18, dtype=np.float32).reshape(6, 3))
    dirty = np.ones(n, np.uint8)
    dirty[roots[2]] = 0
    pos = ob.lights_from_entities(carriers, scene["pos_scale"], scene["parent"], dirty, L)
    exp = L["pos"].copy()
    exp[3] = scene["pos_scale"][roots[3], :3] + carriers["off"][4]     # the later carrier of slot 3 wins
    assert np.array_equal(pos, exp), "released slot, attached entity, clean entity and bad slot are skipped"


# ---------------------------------------------------------------- GPU
def _gpu_tiles(lights, vm, pm, w, h, cell, device):
    from clap_amd import lights as gl
    ls = gl.LightSet(device, w, h, cell)
    ls.load(lights)
    if ls.grid_compute(vm, pm) is None:
        return None
    return ls.download_tiles()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_matches_reference_fixture(name, golden_dir, cuda_device):
    z, lights, w, h, cell = _load(golden_dir, name)
    tiles = _gpu_tiles(lights, z["ref_view_mx"], z["ref_proj_mx"], w, h, cell, cuda_device)
    assert np.array_equal(tiles, z["ref_tiles"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_hip_matches_oracle_random(seed, cuda_device):
    lights, cam, w, h, cell = _random_case(200 + seed)
    _fr, vm, pm = ob.frustum_from_camera(cam)
    exp = ob.light_grid_compute(lights, vm, pm, w, h, cell)
    tiles = _gpu_tiles(lights, vm, pm, w, h, cell, cuda_device)
    assert tiles.shape == exp.shape
    assert np.array_equal(tiles, exp)


@pytest.mark.gpu
def test_hip_4k_grid_and_slot_api(cuda_device):
    """3840x2160 at cell 8 (129 600 tiles); slots driven through the light_get / light_set_* mirror."""
    from clap_amd import lights as gl
    src = synth.lights(128, seed=31)
    ls = gl.LightSet(cuda_device)
    ls.resize(3840, 2160)
    ls.cell = 8
    for i in range(128):
        idx = ls.light_get()
        assert idx == i
        ls.light_set_pos(idx, src["pos"][i])
        ls.light_set_color(idx, src["color"][i])
        ls.light_set_attenuation(idx, src["attenuation"][i])
        ls.light_set_directional(idx, src["is_dir"][i])
    with pytest.raises(Exception):
        ls.light_get()                                      # CERR_TOO_LARGE: all 128 slots taken
    for i in np.flatnonzero(src["active"] == 0):
        ls.light_put(int(i))
    ls.light_set_pos(int(np.flatnonzero(src["active"] == 0)[0]), (9, 9, 9))      # no-op on a released slot
    _fr, vm, pm = ob.frustum_from_camera(synth.camera(pos=(0, 3, 10)))
    ls.grid_compute(vm, pm)
    exp = ob.light_grid_compute(src, vm, pm, 3840, 2160, 8)
    assert np.array_equal(ls.download_tiles(), exp)
    ls.resize(1280, 720)                                    # light_handle_input resize -> new tile counts
    ls.cell = 64
    ls.grid_compute(vm, pm)
    assert np.array_equal(ls.download_tiles(), ob.light_grid_compute(src, vm, pm, 1280, 720, 64))


@pytest.mark.gpu
def test_hip_degenerate_grids_and_errors(cuda_device):
    import ctypes as C
    import torch
    from clap_amd import _lib, lights as gl
    _fr, vm, pm = ob.frustum_from_camera(synth.camera())
    ls = gl.LightSet(cuda_device, 0, 0, 64)
    assert ls.grid_compute(vm, pm) is None                  # light.c:46: no size yet -> nothing happens
    ls = gl.LightSet(cuda_device, 640, 360, 64)             # no lights at all: every mask is zero
    ls.grid_compute(vm, pm)
    assert not ls.download_tiles().any()
    desc = ls._desc()
    desc.nr_lights = 129
    f = (C.c_float * 16)()
    rc = _lib.lib().clapgpu_light_grid_compute(None, C.byref(desc), f, f, 64, 64, 64, ls.tiles.data_ptr())
    assert rc == _lib.ERR_TOO_LARGE
    desc.nr_lights = 1
    rc = _lib.lib().clapgpu_light_grid_compute(None, C.byref(desc), f, f, 64, 64, 64, None)
    assert rc == _lib.ERR_INVALID_ARGUMENTS
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_hip_lights_from_entities(cuda_device):
    from clap_amd import entities, lights as gl, _lib
    scene = synth.pad_levels(synth.entities_chains(40, 3, seed=5))
    L = synth.lights(16, seed=3, inactive_frac=0.0)
    L["active"][5] = 0
    roots = np.flatnonzero(scene["parent"] < 0)[:6]
    child = int(np.flatnonzero(scene["parent"] >= 0)[0])
    carriers = dict(entity=np.asarray([roots[0], roots[1], child, roots[2], roots[3], roots[4]], np.uint32),
                    light=np.asarray([3, 5, 6, 7, 3, 40], np.int32),
                    off=np.arange(18, dtype=np.float32).reshape(6, 3))
    scene["flags"] = scene["flags"].copy()
    scene["flags"][roots[2]] &= ~np.uint32(_lib.E_DIRTY)
    dirty = ((scene["flags"] & _lib.E_DIRTY) != 0).astype(np.uint8)
    batch = entities.EntityBatch(scene, cuda_device)
    ls = gl.LightSet(cuda_device, 640, 360, 64)
    ls.load(L)
    ls.set_carriers(carriers["entity"], carriers["light"], carriers["off"])
    ls.from_entities(batch)
    exp = ob.lights_from_entities(carriers, scene["pos_scale"], scene["parent"], dirty, L)
    assert np.array_equal(ls.download_pos()[:16], exp)
    ls.load(L)
    ls.from_entities(batch, all_dirty=True)                 # CLAPGPU_UPDATE_ALL_DIRTY: the clean carrier applies too
    exp = ob.lights_from_entities(carriers, scene["pos_scale"], scene["parent"], np.ones(scene["n"], np.uint8), L)
    assert np.array_equal(ls.download_pos()[:16], exp)
