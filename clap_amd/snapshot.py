"""ctypes face of the SoA scene snapshot format (include/clapgpu_snapshot.h, libclapgpu_scene.so).

``save`` / ``load`` move a dict of numpy arrays through the C writer / reader; ``save_scene`` /
``load_scene`` group them by component with dotted names ("entities.pos_scale", "camera.persp",
"bodies.pos" ...) -- the arrays ``clap_amd.synth`` produces and the host mirrors consume.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

SCENE_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libclapgpu_scene.so")

DTYPES = {1: np.uint8, 2: np.int32, 3: np.uint32, 4: np.float32, 5: np.float64, 6: np.uint64, 7: np.int64}
CODES = {np.dtype(v): k for k, v in DTYPES.items()}


class _Array(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_uint32), ("ndim", C.c_uint32), ("dims", C.c_uint64 * 4),
                ("count", C.c_uint64), ("data", C.c_void_p)]


# every function include/clapgpu_snapshot.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "clapgpu_dtype_size": (C.c_size_t, [C.c_uint32]),
    "clapgpu_snapshot_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p]),
    "clapgpu_snapshot_add": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p]),
    "clapgpu_snapshot_finish": (C.c_int, [C.c_void_p]),
    "clapgpu_snapshot_abort": (None, [C.c_void_p]),
    "clapgpu_snapshot_open": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p]),
    "clapgpu_snapshot_count": (C.c_uint32, [C.c_void_p]),
    "clapgpu_snapshot_at": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(_Array)]),
    "clapgpu_snapshot_find": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(_Array)]),
    "clapgpu_snapshot_close": (None, [C.c_void_p]),
}
# include/clapgpu_load.h
LOAD_SYMBOLS = {
    "clapgpu_load_scene": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "clapgpu_load_gltf": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
}

_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(SCENE_LIB_PATH):
            raise _lib.ClapGpuError(_lib.ERR_INIT_FAILED, "clap_amd.snapshot", f"{SCENE_LIB_PATH} is not built")
        _lib.lib()                                           # libclapgpu.so first: the scene library links against it
        L = C.CDLL(SCENE_LIB_PATH)
        for name, (res, args) in {**SYMBOLS, **LOAD_SYMBOLS}.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _L = L
    return _L


def save(path, arrays):
    """arrays: {name: ndarray} (u8 / i32 / u32 / f32 / f64 / u64 / i64, up to 4 dims)."""
    L = lib()
    w = C.c_void_p()
    _lib.check(L.clapgpu_snapshot_create(C.byref(w), os.fsencode(path)), "clapgpu_snapshot_create")
    try:
        for name, a in arrays.items():
            a = np.asarray(a)
            if not a.flags.c_contiguous:
                a = a.copy()                                 # (ascontiguousarray would turn 0-d into 1-d)
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            if a.dtype not in CODES or a.ndim > 4:
                raise ValueError(f"{name}: unsupported dtype {a.dtype} or rank {a.ndim}")
            dims = (C.c_uint64 * 4)(*a.shape)
            _lib.check(L.clapgpu_snapshot_add(w, name.encode(), CODES[a.dtype], a.ndim, dims, a.ctypes.data),
                       f"clapgpu_snapshot_add({name})")
    except Exception:
        L.clapgpu_snapshot_abort(w)
        raise
    _lib.check(L.clapgpu_snapshot_finish(w), "clapgpu_snapshot_finish")


def load(path):
    """-> {name: ndarray} (copies: the reader's buffer is released before returning)."""
    L = lib()
    s = C.c_void_p()
    _lib.check(L.clapgpu_snapshot_open(C.byref(s), os.fsencode(path)), f"clapgpu_snapshot_open({path})")
    out = {}
    try:
        for k in range(L.clapgpu_snapshot_count(s)):
            a = _Array()
            _lib.check(L.clapgpu_snapshot_at(s, k, C.byref(a)), "clapgpu_snapshot_at")
            dt = np.dtype(DTYPES[a.dtype])
            shape = tuple(int(a.dims[d]) for d in range(a.ndim))
            if a.count:
                raw = C.string_at(a.data, int(a.count) * dt.itemsize)
                out[a.name.decode()] = np.frombuffer(raw, dt).reshape(shape).copy()
            else:
                out[a.name.decode()] = np.zeros(shape, dt)
    finally:
        L.clapgpu_snapshot_close(s)
    return out


def save_scene(path, **components):
    """components: entities=scene_dict, camera=cam_dict, bodies=..., particles=..., lights=... ;
    ndarray members are stored as "<component>.<key>", scalars as 1-element arrays."""
    arrays = {}
    for comp, d in components.items():
        for k, v in d.items():
            if isinstance(v, (int, np.integer)):
                v = np.asarray([v], np.int64)
            elif isinstance(v, (float, np.floating)):
                v = np.asarray([v], np.float64)
            if isinstance(v, np.ndarray):
                arrays[f"{comp}.{k}"] = v
    save(path, arrays)


def load_scene(path):
    """-> {component: {key: ndarray}}; 1-element i64 / f64 arrays written from scalars come back as scalars."""
    comps = {}
    for name, a in load(path).items():
        comp, _, key = name.partition(".")
        if a.shape == (1,) and a.dtype == np.int64:
            a = int(a[0])
        elif a.shape == (1,) and a.dtype == np.float64:
            a = float(a[0])
        comps.setdefault(comp, {})[key] = a
    return comps


# ---- scene files --------------------------------------------------------------------------------
def load_scene_json(scene_json, snapshot_path, asset_dir=None):
    """clapgpu_load_scene (include/clapgpu_load.h): CLAP's scene.json + glTF assets -> snapshot file."""
    L = lib()
    err = C.create_string_buffer(512)
    rc = L.clapgpu_load_scene(os.fsencode(scene_json), None if asset_dir is None else os.fsencode(asset_dir),
                              os.fsencode(snapshot_path), err, len(err))
    if rc:
        raise _lib.ClapGpuError(rc, "clapgpu_load_scene", err.value.decode(errors="replace"))


def load_gltf(gltf_path, snapshot_path, fix_origin=False):
    L = lib()
    err = C.create_string_buffer(512)
    rc = L.clapgpu_load_gltf(os.fsencode(gltf_path), int(bool(fix_origin)), os.fsencode(snapshot_path), err, len(err))
    if rc:
        raise _lib.ClapGpuError(rc, "clapgpu_load_gltf", err.value.decode(errors="replace"))


def skinned_models(comps):
    """The skinned models of a loaded scene snapshot as the dicts clap_amd.animation.SkinnedModel takes:
    {model index: (skeleton, [animation, ...], mesh)}.  `order` lists the joints reachable from joint 0,
    parents first (the engine's walk starts at joint 0, model.c:1583)."""
    out = {}
    for k in range(int(comps.get("scene", {}).get("n_models", 0))):
        m = comps.get(f"model{k}")
        if not m or not int(m["nr_joints"]):
            continue
        J = int(m["nr_joints"])
        parent = np.asarray(m["joint_parent"], np.int32)
        depth = np.full(J, -1, np.int64)
        depth[0] = 0
        changed = True
        while changed:
            changed = False
            for j in range(1, J):
                if depth[j] < 0 and parent[j] >= 0 and depth[parent[j]] >= 0:
                    depth[j] = depth[parent[j]] + 1
                    changed = True
        reach = np.flatnonzero(depth >= 0)
        order = reach[np.argsort(depth[reach], kind="stable")].astype(np.int32)
        sk = dict(nr_joints=J, parent=parent, invmx=m["invmx"], bind=m["bind"], root_pose=m["root_pose"], order=order,
                  depth=depth.astype(np.int32), joint_types=m["joint_types"])
        anims = []
        for a in range(int(m["n_anims"])):
            an = {key: m[f"a{a}_{key}"] for key in ("ch_target", "ch_path", "ch_nr", "ch_time_off", "ch_data_off", "times", "data")}
            an["n_channels"] = int(an["ch_target"].shape[0])
            an["time_end"] = np.float32(m[f"a{a}_time_end"][0])
            anims.append(an)
        mesh = dict(n_verts=int(m["n_verts"]), position=m["position"],
                    normal=m.get("normal", np.zeros_like(m["position"])), joints=m["joints"], weights=m["weights"])
        out[k] = (sk, anims, mesh)
    return out
