"""Shared helpers of the parity tests."""
import numpy as np

from clap_amd import synth

SCENE_KEYS = ("pos_scale", "rot", "parent", "model", "model_aabb", "model_skip", "flags", "level_start")
CAM_KEYS = ("cam_pos", "cam_quat", "persp", "ndc_z_zero_one")


def bits_equal(a, b):
    """Bit-exact comparison of float arrays (distinguishes -0.0 from 0.0, compares NaN payloads)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def assert_bits_equal(a, b, what):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    bad = np.flatnonzero(a.view(np.uint32).ravel() != b.view(np.uint32).ravel())
    assert bad.size == 0, (f"{what}: {bad.size} of {a.size} floats differ bitwise; first at flat index "
                           f"{bad[0]}: {a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}")


def load_golden(path):
    z = np.load(path)
    scene = {k: z["in_" + k] for k in SCENE_KEYS}
    scene["n"] = scene["parent"].shape[0]
    scene["seqs"] = np.zeros(scene["n"], np.uint32)
    cam = {k: z["in_" + k] for k in CAM_KEYS}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    frames = None
    if "in_frames_pos_scale" in z.files:
        frames = list(zip(z["in_frames_pos_scale"], z["in_frames_rot"], z["in_frames_dirty"]))
    return scene, cam, ref, frames


def apply_frame(scene, st, frame):
    """Host-side mutation a game would do between frames: write TRS of dirty entities, set dirty."""
    ps, rot, dirty = frame
    idx = np.flatnonzero(dirty)
    scene["pos_scale"][idx] = ps[idx]
    scene["rot"][idx] = rot[idx]
    st["flags"][idx] |= synth.E_DIRTY
    return idx


# ---- the 1e-5 bar of the floating-point rows, held per object rather than per array ----
# north_star: "within 1e-5 relative for float transforms/positions".  Relative to WHAT decides how hard that is: against
# the largest magnitude of a whole array, rotation entries (<= 1) of a palette whose translations reach 10^2..10^3 would
# only be held to 1e-3..1e-2 absolute.  Here every object carries its own scale: each joint's 3x3 block, each joint's
# translation column, each T, R and S, each joint position, each vertex position and normal --
# max |got - ref| over the object / max |ref| over the SAME object (floor 1e-30), bar 1e-5.
RTOL = 1e-5
MAT3_IDX = (0, 1, 2, 4, 5, 6, 8, 9, 10)          # column-major mat4: the linear block
TCOL_IDX = (12, 13, 14)                          # its translation column
BROW_IDX = (3, 7, 11, 15)                        # bottom row: (0, 0, 0, 1) of an affine matrix, held absolutely
PARITY_BOUNDS = {}                               # worst case seen per quantity; conftest dumps it at session end


def _rel_objects(got, exp, idx=None):
    g = np.asarray(got, np.float64)
    e = np.asarray(exp, np.float64)
    assert g.shape == e.shape, (g.shape, e.shape)
    if idx is not None:
        g, e = g[..., list(idx)], e[..., list(idx)]
    d = np.abs(g - e).max(axis=-1)
    s = np.abs(e).max(axis=-1)
    r = d / np.maximum(s, 1e-30)
    r[~np.isfinite(d)] = np.inf                  # a NaN / inf on either side is a failure, not a pass
    return r


COND_ULPS = 64                                   # see _hold(): the forward-error allowance of a cancelled sum, in fp32 ulps of its terms
U32 = 2.0 ** -24


def _hold(r, what, tol=RTOL, err=None, terms=None):
    """r: per-object relative error.  `err` / `terms` (same shape, optional): the object's absolute error and the magnitude of
    the largest TERM of the sum it is.  A sum that cancels (|result| << |terms|) carries the rounding of its terms --
    about one fp32 ulp of them per operation, in ANY evaluation order but the reference's very own, the reference's
    own NEON path (linmath.h:459-504) included -- so relative to its own magnitude it cannot be held to 1e-5 by
    construction.  Such an object passes on the forward-error bound |err| <= COND_ULPS * 2^-24 * terms instead; how many
    objects needed that, and the worst of them in ulps of their terms, is recorded next to the plain worst case."""
    r = np.asarray(r)
    over = r > tol if np.all(np.isfinite(r)) else ~(r <= tol)
    if terms is not None and over.any():
        ulps = np.asarray(err)[over] / np.maximum(U32 * np.asarray(terms)[over], 1e-300)
        PARITY_BOUNDS[what + " | objects over 1e-5 of their own magnitude (cancelled sums)"] = \
            PARITY_BOUNDS.get(what + " | objects over 1e-5 of their own magnitude (cancelled sums)", 0) + int(over.sum())
        k2 = what + " | ... their worst error in fp32 ulps of their terms (allowed: %d)" % COND_ULPS
        PARITY_BOUNDS[k2] = max(PARITY_BOUNDS.get(k2, 0.0), float(ulps.max()))
        k3 = what + " | ... their largest magnitude relative to their terms"
        PARITY_BOUNDS[k3] = max(PARITY_BOUNDS.get(k3, 0.0), float((np.asarray(err)[over] / np.maximum(r[over], 1e-300)
                                                                     / np.maximum(np.asarray(terms)[over], 1e-300)).max()))
        ok = ulps <= COND_ULPS
        if not ok.all():
            raise AssertionError(f"{what}: {int((~ok).sum())} objects are off by more than 1e-5 of their own magnitude AND by "
                                 f"more than {COND_ULPS} ulp of their terms (worst {float(ulps.max()):.1f} ulp)")
        r = np.where(over, 0.0, r)
    worst = float(r.max()) if r.size else 0.0
    PARITY_BOUNDS[what] = max(PARITY_BOUNDS.get(what, 0.0), worst)
    if not worst <= tol:
        at = np.unravel_index(int(np.argmax(r)), r.shape)
        raise AssertionError(f"{what}: worst object {at} off by {worst:.3e} of its own magnitude (bar {tol:g}); "
                             f"{int((r > tol).sum())} of {r.size} objects over the bar")
    return worst


def _abs_err(got, exp, idx=None):
    g = np.asarray(got, np.float64)
    e = np.asarray(exp, np.float64)
    if idx is not None:
        g, e = g[..., list(idx)], e[..., list(idx)]
    return np.abs(g - e).max(axis=-1)


def assert_vec_close(got, exp, what, key=None, terms=None):
    """(..., k) arrays of k-vectors (positions, normals, quaternions): each vector against its own magnitude.
    terms: per-vector magnitude of the largest term of the sum the vector is (see _hold)."""
    return _hold(_rel_objects(got, exp), what if key is None else key,
                 err=None if terms is None else _abs_err(got, exp), terms=terms)


def assert_mat4_close(got, exp, what, key=None, t_terms=None):
    """(..., 16) column-major affine matrices: linear block and translation column separately, each per matrix.
    t_terms: per-matrix magnitude of the largest term of the translation column's sum (see _hold)."""
    key = what if key is None else key
    a = _hold(_rel_objects(got, exp, MAT3_IDX), f"{what} [3x3 block]" if key is what else key + " [3x3 block]")
    b = _hold(_rel_objects(got, exp, TCOL_IDX), f"{what} [translation]" if key is what else key + " [translation]",
              err=None if t_terms is None else _abs_err(got, exp, TCOL_IDX), terms=t_terms)
    g = np.asarray(got, np.float64)[..., list(BROW_IDX)]
    e = np.asarray(exp, np.float64)[..., list(BROW_IDX)]
    assert float(np.abs(g - e).max(initial=0.0)) <= RTOL, f"{what}: bottom row"
    return max(a, b)


def assert_trs_close(got, exp, what, key=None):
    """(..., 10) joint T(3) R(4, xyzw) S(3): each of the three against its own magnitude."""
    key = what if key is None else key
    return max(_hold(_rel_objects(got, exp, (0, 1, 2)), key + " [T]"),
               _hold(_rel_objects(got, exp, (3, 4, 5, 6)), key + " [R]"),
               _hold(_rel_objects(got, exp, (7, 8, 9)), key + " [S]"))


# ---- magnitudes of the terms behind the summed quantities of the pose path (numpy, from the ORACLE's results) ----
def _abs3(m16):
    """|linear block| of column-major mat4s as (..., row, col)."""
    m = np.abs(np.asarray(m16, np.float64))
    return np.stack([m[..., [0, 4, 8]], m[..., [1, 5, 9]], m[..., [2, 6, 10]]], axis=-2)


def pose_term_scales(sk, gl, jt, char_mx, entity_rows=None):
    """Per (character, joint): the largest term magnitude behind (a) the palette's translation column
    joint_transforms = global * invmx  (model.c:1389: t = G3 * t_inv + t_G, with t_G itself the sum of the rotated local
    translations down the joint's ancestor path, model.c:1363-1383) and (b) the joint's world position
    e->mx * (joint_transforms * bind) * (0,0,0,1)  (model.c:1392-1400)."""
    gl = np.asarray(gl, np.float64)
    n, J = gl.shape[0], gl.shape[1]
    parent = np.asarray(sk["parent"])
    inv = np.asarray(sk["invmx"], np.float64).reshape(J, 16)
    bind = np.asarray(sk["bind"], np.float64).reshape(J, 16)
    tG = np.abs(gl[..., 12:15]).max(axis=-1)                                  # (n, J)
    chain = tG.copy()
    for j in np.asarray(sk["order"]):                                         # parents first
        if parent[j] >= 0:
            chain[:, j] = np.maximum(chain[:, j], chain[:, parent[j]])
    chain = np.maximum(chain, np.abs(np.asarray(sk["root_pose"], np.float64).reshape(16)[12:15]).max())
    rot_t = np.einsum("njrc,jc->njr", _abs3(gl), np.abs(inv[:, 12:15])).max(axis=-1)
    s_jt = np.maximum(chain, rot_t)                                           # (n, J)
    s_mpos = np.maximum(s_jt, np.einsum("njrc,jc->njr", _abs3(jt), np.abs(bind[:, 12:15])).max(axis=-1))
    em = np.asarray(char_mx, np.float64).reshape(-1, 16)
    if entity_rows is not None:
        em = em[np.asarray(entity_rows)]
    e3 = _abs3(em).sum(axis=-1).max(axis=-1)                                  # (n,): largest absolute row sum
    s_pos = np.maximum(e3[:, None] * s_mpos, np.abs(em[:, 12:15]).max(axis=-1)[:, None])
    return s_jt, s_pos


def skin_term_scales(mesh, vert_first, vert_count, jt, jt_terms, chunk=1 << 20):
    """Per output vertex: the largest |w_i| * (|J3_i| |p| + |t_i| term scale) over its four influences (model.vert:35-42)."""
    out = np.zeros(int(np.sum(vert_count)), np.float64)
    J = jt.shape[1]
    a3 = _abs3(jt)                                                            # (n, J, 3, 3)
    at = 0
    for c in range(len(vert_count)):
        f, k = int(vert_first[c]), int(vert_count[c])
        p = np.abs(np.asarray(mesh["position"][f:f + k], np.float64))
        jj = np.asarray(mesh["joints"][f:f + k]).astype(np.int64)
        w = np.abs(np.asarray(mesh["weights"][f:f + k], np.float64))
        rot = np.einsum("virc,vc->vir", a3[c][jj], p).max(axis=-1)              # (k, 4)
        out[at:at + k] = (w * np.maximum(rot, jt_terms[c][jj])).max(axis=-1)
        at += k
    return out
