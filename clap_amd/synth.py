"""Deterministic synthetic scenes for tests and bench.py (numpy only).

The reference ships no scene fixtures (its asset submodules are empty), so the
configurations of BASELINE.json are reconstructed from the loader's schema and
the instantiator's parameter ranges as SURVEY.md section 8(d) specifies.  All
randomness comes from numpy's PCG64 with fixed seeds -- never libc rand.

Array conventions are the device SoA layout of include/clapgpu.h:
``pos_scale[n,4] = (x, y, z, scale)``, ``rot[n,4] = quat (x, y, z, w)``,
``parent[n]`` (-1 = root, otherwise ``parent[i] < i``), ``model[n]``,
``model_aabb[m,6] = (min xyz, max xyz)``, ``flags[n]``, ``seqs[n]``.
"""
import math

import numpy as np

E_VISIBLE = np.uint32(1 << 0)
E_SKIP_CULLING = np.uint32(1 << 14)
E_DIRTY = np.uint32(1 << 16)
E_ALIVE = np.uint32(1 << 31)

F32 = np.float32


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def quat_from_euler_xyz(x, y, z):
    """Same formula as the reference's quat_from_euler_xyz (linmath.h:857-870), fp32."""
    x, y, z = (np.asarray(v, dtype=F32) for v in (x, y, z))
    h = F32(0.5)
    cx, sx = np.cos(x * h, dtype=F32), np.sin(x * h, dtype=F32)
    cy, sy = np.cos(y * h, dtype=F32), np.sin(y * h, dtype=F32)
    cz, sz = np.cos(z * h, dtype=F32), np.sin(z * h, dtype=F32)
    q = np.stack([sx * cy * cz - cx * sy * sz,
                  cx * sy * cz + sx * cy * sz,
                  cx * cy * sz - sx * sy * cz,
                  cx * cy * cz + sx * sy * sz], axis=-1)
    return q.astype(F32)


def camera(pos=(0.0, 0.0, 0.0), quat=(0.0, 0.0, 0.0, 1.0), fov_deg=70.0, aspect=16.0 / 9.0,
           near=0.1, far=500.0, ndc_z_zero_one=0):
    """Default scene camera (scene.c:74-76): fov 70 deg, near 0.1, far 500, looking down -Z."""
    return dict(cam_pos=np.asarray(pos, F32), cam_quat=np.asarray(quat, F32),
                persp=np.asarray([F32(fov_deg) * F32(math.pi) / F32(180.0), aspect, near, far], F32),
                ndc_z_zero_one=np.asarray([ndc_z_zero_one], np.uint32))


def _models_default():
    return dict(model_aabb=np.asarray([[-1, -2, -3, 1, 2, 3]], F32),
                model_skip=np.zeros(1, np.uint8))


def _pack(pos, scale, rot, parent, model=None, flags=None):
    n = pos.shape[0]
    d = dict(n=n,
             pos_scale=np.concatenate([pos, scale[:, None]], axis=1).astype(F32),
             rot=rot.astype(F32),
             parent=parent.astype(np.int32),
             model=np.zeros(n, np.int32) if model is None else model.astype(np.int32),
             flags=(np.full(n, E_ALIVE | E_VISIBLE | E_DIRTY, np.uint32) if flags is None else flags),
             seqs=np.zeros(n, np.uint32))
    d.update(_models_default())
    return d


def _root_population(rng, n, full_euler):
    """Instantiator population (terrain.c:559-568 -> model.c:1863-1874): random yaw,
    scale 1 + 0.5 (1 - 2u); positions U(-500,500) x U(-50,50) x U(-500,500)."""
    pos = np.stack([rng.uniform(-500, 500, n), rng.uniform(-50, 50, n), rng.uniform(-500, 500, n)], 1)
    scale = 1.0 + 0.5 * (1.0 - 2.0 * rng.uniform(0, 1, n))
    if full_euler:
        ang = rng.uniform(-math.pi, math.pi, (n, 3))
        rot = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2])
    else:
        ry = rng.uniform(0, 2 * math.pi, n)
        rot = quat_from_euler_xyz(np.zeros(n), ry, np.zeros(n))
    return pos.astype(F32), scale.astype(F32), rot


def entities_flat(n=10_000, seed=1234, full_euler=False):
    """C1: flat (parentless) entities, one shared model AABB (-1,-2,-3)..(1,2,3)."""
    rng = _rng(seed)
    pos, scale, rot = _root_population(rng, n, full_euler)
    return _pack(pos, scale, rot, np.full(n, -1))


def entities_chains(n_chains=125_000, depth=8, seed=2):
    """C2: n_chains x depth entities, level-major (entity = level * n_chains + chain).
    Roots as C1; children local pos U(-2,2)^3, scale U(0.8,1.2), random full rotation."""
    rng = _rng(seed)
    n = n_chains * depth
    pos = np.empty((n, 3), F32)
    scale = np.empty(n, F32)
    rot = np.empty((n, 4), F32)
    parent = np.empty(n, np.int64)
    p0, s0, r0 = _root_population(rng, n_chains, False)
    pos[:n_chains], scale[:n_chains], rot[:n_chains] = p0, s0, r0
    parent[:n_chains] = -1
    for lvl in range(1, depth):
        a, b = lvl * n_chains, (lvl + 1) * n_chains
        pos[a:b] = rng.uniform(-2, 2, (n_chains, 3))
        scale[a:b] = rng.uniform(0.8, 1.2, n_chains)
        ang = rng.uniform(-math.pi, math.pi, (n_chains, 3))
        rot[a:b] = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2])
        parent[a:b] = np.arange(a - n_chains, b - n_chains)
    d = _pack(pos, scale, rot, parent)
    d["level_start"] = np.arange(0, n + 1, n_chains, dtype=np.uint32)
    return d


def entities_forest(n=5000, seed=7, max_depth=6, n_models=3, dead_frac=0.05, hidden_frac=0.05,
                    skipcull_frac=0.05):
    """Ragged random forest for parity tests: random fan-out, several models (one with
    skip_aabb), dead / hidden / skip-culling entities.  Stored level-major."""
    rng = _rng(seed)
    depth = np.zeros(n, np.int64)
    parent = np.full(n, -1, np.int64)
    n_roots = max(1, n // 4)
    for i in range(n_roots, n):
        p = int(rng.integers(0, i))
        if depth[p] + 1 >= max_depth:
            p = int(rng.integers(0, n_roots))
        parent[i] = p
        depth[i] = depth[p] + 1
    order = np.argsort(depth, kind="stable")          # level-major, parents first
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    parent = np.where(parent[order] >= 0, inv[np.maximum(parent[order], 0)], -1)
    depth = depth[order]
    pos_r, scale_r, rot_r = _root_population(rng, n, True)
    is_child = parent >= 0
    pos = np.where(is_child[:, None], rng.uniform(-3, 3, (n, 3)), pos_r).astype(F32)
    scale = np.where(is_child, rng.uniform(0.7, 1.3, n), scale_r).astype(F32)
    flags = np.full(n, E_ALIVE | E_VISIBLE | E_DIRTY, np.uint32)
    u = rng.uniform(0, 1, n)
    flags[u < dead_frac] &= ~E_ALIVE
    flags[(u >= dead_frac) & (u < dead_frac + hidden_frac)] &= ~E_VISIBLE
    flags[(u >= 1 - skipcull_frac)] |= E_SKIP_CULLING
    model = rng.integers(0, n_models, n)
    d = _pack(pos, scale, rot_r, parent, model=model, flags=flags)
    aabb_lo = -rng.uniform(0.5, 3.0, (n_models, 3))
    aabb_hi = rng.uniform(0.5, 3.0, (n_models, 3))
    d["model_aabb"] = np.concatenate([aabb_lo, aabb_hi], 1).astype(F32)
    d["model_skip"] = np.zeros(n_models, np.uint8)
    if n_models > 1:
        d["model_skip"][n_models - 1] = 1
    counts = np.bincount(depth, minlength=int(depth.max()) + 1)
    d["level_start"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    return d


def level_starts(parent):
    """Level boundaries of a level-major parent array (parents strictly before children)."""
    parent = np.asarray(parent)
    n = parent.shape[0]
    depth = np.zeros(n, np.int64)
    for i in range(n):
        if parent[i] >= 0:
            depth[i] = depth[parent[i]] + 1
    assert np.all(np.diff(depth) >= 0), "entities are not stored level-major"
    counts = np.bincount(depth, minlength=int(depth.max()) + 1 if n else 1)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)


def pad_levels(scene, align=64):
    """Re-lay an entity scene so every hierarchy level starts at a multiple of `align`
    (include/clapgpu.h: one wavefront owns one vis_mask word).  Padding slots are dead
    entities (flags == 0).  Adds ``slot_of`` (original -> padded index) and
    ``orig_of`` (padded -> original index, -1 for padding)."""
    ls = scene.get("level_start")
    if ls is None:
        ls = level_starts(scene["parent"])
    ls = np.asarray(ls, np.int64)
    n = int(scene["n"])
    slot_of = np.empty(n, np.int64)
    new_ls = [0]
    for l in range(len(ls) - 1):
        a, b = int(ls[l]), int(ls[l + 1])
        start = new_ls[-1]
        slot_of[a:b] = start + np.arange(b - a)
        end = start + (b - a)
        new_ls.append(end if l == len(ls) - 2 else -(-end // align) * align)
    n_pad = new_ls[-1]
    orig_of = np.full(n_pad, -1, np.int64)
    orig_of[slot_of] = np.arange(n)

    def scatter(a, fill=0):
        out = np.full((n_pad,) + a.shape[1:], fill, a.dtype)
        out[slot_of] = a
        return out

    out = dict(scene)
    out["n"] = n_pad
    out["n_real"] = n
    out["pos_scale"] = scatter(scene["pos_scale"])
    out["pos_scale"][orig_of < 0, 3] = 1.0
    out["rot"] = scatter(scene["rot"])
    out["rot"][orig_of < 0, 3] = 1.0
    par = scene["parent"].astype(np.int64)
    out["parent"] = scatter(np.where(par >= 0, slot_of[np.maximum(par, 0)], -1).astype(np.int32), -1)
    out["model"] = scatter(scene["model"])
    out["flags"] = scatter(scene["flags"])
    out["seqs"] = scatter(scene["seqs"])
    out["level_start"] = np.asarray(new_ls, np.uint32)
    out["slot_of"] = slot_of
    out["orig_of"] = orig_of
    return out


def model_table(scene):
    """Device model table float32[m][8]: (min.xyz, skip_aabb as uint32 bits, max.xyz, 0)."""
    m = scene["model_aabb"].shape[0]
    t = np.zeros((m, 8), np.float32)
    t[:, 0:3] = scene["model_aabb"][:, 0:3]
    t[:, 4:7] = scene["model_aabb"][:, 3:6]
    t.view(np.uint32)[:, 3] = scene["model_skip"].astype(np.uint32)
    if "model_lod" in scene:                                  # (lod_min, lod_max) per model
        ml = np.asarray(scene["model_lod"], np.uint32)
        t.view(np.uint32)[:, 7] = ml[:, 0] | (ml[:, 1] << 8)
    return t


# ----------------------------------------------------------------------------- particles
PART_DIST_LIN, PART_DIST_SQRT, PART_DIST_CBRT, PART_DIST_POW075 = 0, 1, 2, 3
DRAND48_DEFAULT_STATE = 0x1234ABCD330E      # glibc's initial drand48 state (the reference never seeds it)

PSYS_DTYPE = np.dtype([("center", np.float32, 3), ("dist", np.uint32),
                       ("radius", np.float64), ("min_radius", np.float64),
                       ("radius_squared", np.float64), ("velocity", np.float64),
                       ("first", np.uint32), ("count", np.uint32), ("pad", np.uint32, 2)])
assert PSYS_DTYPE.itemsize == 64


def particle_systems(n_sys=4096, count=1024, radius=10.0, min_radius=0.0, velocity=0.005,
                     dist=PART_DIST_SQRT, seed=4, ragged=False):
    """C4 (particles part): n_sys systems x count particles, radius 10, velocity 0.005
    (particle.c:223 default), PART_DIST_SQRT.  Every system starts at a multiple of 64
    particles (one wavefront row belongs to one system); `ragged` varies counts, radii and
    distributions for parity tests."""
    rng = _rng(seed)
    sys = np.zeros(n_sys, PSYS_DTYPE)
    sys["center"] = rng.uniform(-200, 200, (n_sys, 3)).astype(F32)
    if ragged:
        sys["count"] = rng.integers(1, count + 1, n_sys)
        sys["radius"] = rng.uniform(0.5, radius, n_sys)
        sys["min_radius"] = sys["radius"] * rng.uniform(0, 0.9, n_sys) * (rng.uniform(0, 1, n_sys) < 0.5)
        sys["velocity"] = rng.uniform(0.5, 2.0, n_sys) * velocity
        sys["dist"] = rng.integers(0, 2, n_sys)          # LIN / SQRT: the bit-exact distributions
    else:
        sys["count"] = count
        sys["radius"] = radius
        sys["min_radius"] = min_radius
        sys["velocity"] = velocity
        sys["dist"] = dist
    sys["radius_squared"] = sys["radius"] * sys["radius"]
    rows = (sys["count"].astype(np.int64) + 63) // 64
    first_row = np.concatenate([[0], np.cumsum(rows)])
    sys["first"] = (first_row[:-1] * 64).astype(np.uint32)
    n = int(first_row[-1]) * 64
    row_sys = np.repeat(np.arange(n_sys, dtype=np.uint32), rows)
    return dict(sys=sys, n=n, n_real=int(sys["count"].sum()), row_sys=row_sys)


_R48_A, _R48_C, _M24 = 0x5DEECE66D, 0xB, (1 << 24) - 1


def _mul48(a, b):
    """a * b mod 2^48 on uint64 arrays (24-bit limbs: no product exceeds 2^48)."""
    a_lo, a_hi, b_lo, b_hi = a & _M24, a >> np.uint64(24), b & _M24, b >> np.uint64(24)
    mid = (a_hi * b_lo + a_lo * b_hi) & np.uint64(_M24)
    return (a_lo * b_lo + (mid << np.uint64(24))) & np.uint64((1 << 48) - 1)


def drand48_stream(state, count):
    """The next `count` values of glibc's drand48 from `state` (X' = (0x5DEECE66D X + 0xB) mod 2^48, value = X' / 2^48)
    and the state after them, by jump-ahead: the maps of 1..B steps (A_i, C_i) are built by doubling once, the block
    starts X_{kB} follow one another through the B-step map, and X_{kB+i} = A_i X_{kB} + C_i for all of them at once."""
    mask = np.uint64((1 << 48) - 1)
    if count <= 0:
        return np.zeros(0, np.float64), int(state)
    B = 1 << min(16, max(0, int(count - 1).bit_length()))
    A = np.asarray([_R48_A], np.uint64)
    Cc = np.asarray([_R48_C], np.uint64)
    while A.shape[0] < B:                                  # (A, C)[m + i] = (A[m-1] A[i], A[i] C[m-1] + C[i])
        a_m, c_m = A[-1], Cc[-1]
        A, Cc = (np.concatenate([A, _mul48(A, np.full_like(A, a_m))]),
                 np.concatenate([Cc, (_mul48(A, np.full_like(A, c_m)) + Cc) & mask]))
    n_blocks = (count + B - 1) // B
    starts = np.empty(n_blocks, np.uint64)
    x = int(state)
    a_b, c_b = int(A[B - 1]), int(Cc[B - 1])
    for k in range(n_blocks):
        starts[k] = x
        x = (a_b * x + c_b) & ((1 << 48) - 1)
    xs = (_mul48(np.broadcast_to(A[None, :B], (n_blocks, B)), np.broadcast_to(starts[:, None], (n_blocks, B)))
          + Cc[None, :B]) & mask
    xs = xs.reshape(-1)[:count]
    return xs.astype(np.float64) * (1.0 / 281474976710656.0), int(xs[-1])


def particles_spawn(ps, rng_state):
    """particle_system_make's spawn loop (particle.c:36-87, 229-234) for every system in order on one drand48 stream:
    per particle a point in the system's sphere (3 draws for the direction, 1 for the radius under the system's
    distribution) and a velocity (3 draws).  Returns (pos[n,3], vel[n,3], stream state afterwards).  Harness-side
    preparation of the device arrays (the engine spawns on the host as well)."""
    sys = ps["sys"]
    n = int(ps["n"])
    pos, vel = np.zeros((n, 3), F32), np.zeros((n, 3), F32)
    counts = sys["count"].astype(np.int64)
    total = int(counts.sum())
    if not total:
        return pos, vel, int(rng_state)
    draws, state = drand48_stream(rng_state, 7 * total)
    d = draws.reshape(total, 7)
    which = np.repeat(np.arange(sys.shape[0]), counts)
    idx = np.repeat(sys["first"].astype(np.int64), counts) + (np.arange(total) - np.repeat(np.cumsum(counts) - counts, counts))
    dirv = (d[:, 0:3] * 2.0 - 1.0).astype(F32)
    dot = F32(0) + dirv[:, 0] * dirv[:, 0]                  # lm_dot3: p = 0; p += b[i] * a[i]
    dot = dot + dirv[:, 1] * dirv[:, 1]
    dot = dot + dirv[:, 2] * dirv[:, 2]
    ln = np.sqrt(dot)
    with np.errstate(divide="ignore"):
        k = (1.0 / ln.astype(np.float64)).astype(F32)       # vec3_norm: double divide, float store
    dirv = np.where((ln != 0)[:, None], dirv * k[:, None], dirv)
    u = d[:, 3].copy()
    dist = sys["dist"][which]
    u = np.where(dist == PART_DIST_SQRT, np.sqrt(d[:, 3]), u)
    u = np.where(dist == PART_DIST_CBRT, np.cbrt(d[:, 3]), u)
    u = np.where(dist == PART_DIST_POW075, np.power(d[:, 3], 0.75), u)
    r = (sys["min_radius"][which] + (sys["radius"][which] - sys["min_radius"][which]) * u).astype(F32)
    pos[idx] = sys["center"][which] * F32(1.0) + dirv * r[:, None]
    vel[idx] = ((d[:, 4:7] * 2.0 - 1.0) * sys["velocity"][which][:, None]).astype(F32)
    return pos, vel, state


# ----------------------------------------------------------------------------- skeletons / animation / meshes
def _rigid_mat4(rng, n, spread=1.0):
    """n random rigid transforms as column-major mat4 [n,16]."""
    ang = rng.uniform(-math.pi, math.pi, (n, 3))
    q = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2]).astype(np.float64)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    m = np.zeros((n, 4, 4))                       # m[i, col, row]
    m[:, 0, 0] = 1 - 2 * (y * y + z * z); m[:, 0, 1] = 2 * (x * y + w * z); m[:, 0, 2] = 2 * (x * z - w * y)
    m[:, 1, 0] = 2 * (x * y - w * z); m[:, 1, 1] = 1 - 2 * (x * x + z * z); m[:, 1, 2] = 2 * (y * z + w * x)
    m[:, 2, 0] = 2 * (x * z + w * y); m[:, 2, 1] = 2 * (y * z - w * x); m[:, 2, 2] = 1 - 2 * (x * x + y * y)
    m[:, 3, :3] = rng.uniform(-spread, spread, (n, 3))
    m[:, 3, 3] = 1
    return m.reshape(n, 16).astype(F32)


def skeleton(nr_joints=64, max_depth=8, seed=3, unreachable=0):
    """One skinned model (C3): binary-ish joint tree of depth <= max_depth rooted at joint 0,
    invmx = inverse of random rigid bind matrices, a rigid root_pose.  `unreachable` extra
    joints hang off a second root (the reference only ever updates joints reachable from
    joint 0, model.c:1583)."""
    rng = _rng(seed)
    J = nr_joints
    parent = np.full(J, -1, np.int32)
    depth = np.zeros(J, np.int64)
    n_main = J - unreachable
    for j in range(1, n_main):
        for _try in range(64):
            p = int(rng.integers(max(0, (j - 1) // 2 - 2), j))
            if depth[p] + 1 < max_depth:
                break
        else:                                            # the window only holds leaves at the depth limit
            p = int(np.flatnonzero(depth[:j] + 1 < max_depth)[-1])
        parent[j], depth[j] = p, depth[p] + 1
    for j in range(n_main, J):
        parent[j] = -1 if j == n_main else j - 1
    bind_world = _rigid_mat4(rng, J, 1.5).astype(np.float64).reshape(J, 4, 4)
    invmx = np.stack([np.linalg.inv(b.T).T for b in bind_world]).reshape(J, 16).astype(F32)
    root_pose = _rigid_mat4(rng, 1, 0.5)[0]
    order = [j for j in np.argsort(depth[:n_main], kind="stable")]       # parents first, joints reachable from 0
    return dict(nr_joints=J, parent=parent, invmx=invmx, root_pose=root_pose,
                order=np.asarray(order, np.int32), depth=depth.astype(np.int32))


def animation(nr_joints=64, keyframes=30, time_end=2.0, seed=3, missing_frac=0.0, ragged=False):
    """One animation: T, R, S channels per joint (3*J channels), `keyframes` strictly increasing
    key times on [0, time_end] each.  `missing_frac` drops channels, `ragged` varies key counts."""
    rng = _rng(seed + 1000)
    tgt, path, nr, toff, doff, times, data = [], [], [], [], [], [], []
    t_at = d_at = 0
    for j in range(nr_joints):
        for p in range(3):
            if rng.uniform() < missing_frac:
                continue
            k = int(rng.integers(2, keyframes + 1)) if ragged else keyframes
            t = np.sort(rng.uniform(0, time_end, k)).astype(F32)
            t[0], t[-1] = 0.0, time_end
            t = np.unique(t)
            k = t.shape[0]
            if p == 1:
                ang = rng.uniform(-1.2, 1.2, (k, 3))
                d = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2])
                flip = rng.uniform(0, 1, k) < 0.3                       # exercises the dot < 0 branch
                d[flip] = -d[flip]
                near = rng.uniform(0, 1, k) < 0.2                       # exercises the nlerp branch (dot > 0.9995)
                d[1:][near[1:]] = d[:-1][near[1:]]
            elif p == 0:
                d = rng.uniform(-0.5, 0.5, (k, 3)).astype(F32)
            else:
                d = rng.uniform(0.8, 1.25, (k, 3)).astype(F32)
            tgt.append(j); path.append(p); nr.append(k); toff.append(t_at); doff.append(d_at)
            times.append(t); data.append(d.astype(F32).ravel())
            t_at += k
            d_at += d.size
    return dict(n_channels=len(tgt), ch_target=np.asarray(tgt, np.uint32), ch_path=np.asarray(path, np.uint32),
                ch_nr=np.asarray(nr, np.uint32), ch_time_off=np.asarray(toff, np.uint32),
                ch_data_off=np.asarray(doff, np.uint32), times=np.concatenate(times).astype(F32),
                data=np.concatenate(data).astype(F32), time_end=F32(time_end))


def characters(n_chars=50_000, nr_joints=64, time_end=2.0, seed=3):
    """Per-character inputs: entity world matrix (rigid, as the entity kernel produces) and an
    animation phase; trs0 = the joints' rest T/R/S (used by joints no channel targets)."""
    rng = _rng(seed + 2000)
    mx = _rigid_mat4(rng, n_chars, 300.0)
    phase = rng.uniform(0, time_end, n_chars).astype(F32)
    trs0 = np.zeros((nr_joints, 10), F32)
    trs0[:, 0:3] = rng.uniform(-0.3, 0.3, (nr_joints, 3))
    ang = rng.uniform(-1, 1, (nr_joints, 3))
    trs0[:, 3:7] = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2])
    trs0[:, 7:10] = 1.0
    return dict(n_chars=n_chars, char_mx=mx, phase=phase, trs0=trs0)


def skinned_mesh(n_verts=200, nr_joints=64, seed=3, copies=1):
    """A skinned mesh in the reference's vertex layout (mesh.h:125-131, gltf.c:387-388):
    position f32x3, normal f32x3, joints u8x4, weights f32x4 (Dirichlet(4): sum to 1)."""
    rng = _rng(seed + 3000)
    n = n_verts * copies
    nor = rng.normal(size=(n, 3))
    nor /= np.linalg.norm(nor, axis=1, keepdims=True)
    w = rng.dirichlet(np.ones(4), n)
    return dict(n_verts=n, position=rng.uniform(-1, 1, (n, 3)).astype(F32), normal=nor.astype(F32),
                joints=rng.integers(0, nr_joints, (n, 4)).astype(np.uint8), weights=w.astype(F32))


# ----------------------------------------------------------------------------- rigid bodies
def sphere_bodies(n=262_144, box=64.0, rmin=0.1, rmax=0.5, seed=4, entity_base=0, resting_frac=0.0):
    """C4 (body half): n sphere bodies, radius U(rmin,rmax), positions uniform in a box^3 cube,
    lvel N(0,1).  SURVEY.md 8d names a 512^3 box "(a few pairs/body)", but 256k spheres of
    diameter <= 1 in 512^3 give ~0.002 pairs/body; box = 64 gives the stated ~1 pair/body and
    is the default.  fp64 state in ODE's layout (quat = w,x,y,z)."""
    rng = _rng(seed)
    radius = rng.uniform(rmin, rmax, n)
    ang = rng.uniform(-math.pi, math.pi, (n, 3))
    q = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2]).astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    lvel = rng.normal(0, 1, (n, 3))
    avel = rng.normal(0, 0.5, (n, 3))
    rest = rng.uniform(0, 1, n) < resting_frac
    lvel[rest] *= 1e-3
    avel[rest] *= 1e-3
    bflags = np.full(n, 2, np.uint32)                        # auto-disable on (physics.c:1039)
    bflags[rest] |= 4                                        # resting bodies: no gravity, so they stay idle
    return dict(n=n, pos=rng.uniform(0, box, (n, 3)), quat=q[:, [3, 0, 1, 2]].copy(), lvel=lvel, avel=avel,
                mass=4.0 / 3.0 * math.pi * radius ** 3, radius=radius, yoffset=radius.copy(),
                bflags=bflags, adis_steps_left=np.full(n, 30, np.int32), adis_time_left=np.zeros(n),
                body_entity=(entity_base + np.arange(n)).astype(np.int32), cell=2.0 * rmax)


def capsule_bodies(n=262_144, box=64.0, seed=4, sphere_frac=0.3, entity_base=0, resting_frac=0.0):
    """C4 with the reference's body geoms: every body goes through phys_geom_capsule_new (physics.c:814-873) on a
    random entity AABB -- upright capsules, "puppy" capsules along Z, and spheres where the capsule length comes
    out 0 (`sphere_frac` of the bodies get a cube AABB, which does exactly that) -- with dMassSetCapsuleTotal /
    dMassSetSphereTotal inertia, random orientations and spins, gyroscopic mode on (dBodyCreate's default)."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()                                          # host helpers only: no GPU needed
    rng = _rng(seed)
    ext = np.empty((n, 3), np.float32)
    kind = rng.uniform(0, 1, n)
    ext[:, 0] = rng.uniform(0.2, 0.5, n)
    ext[:, 1] = rng.uniform(0.9, 1.8, n)                     # tall: upright capsule
    ext[:, 2] = rng.uniform(0.2, 0.5, n)
    puppy = kind > 0.75
    ext[puppy, 1] = rng.uniform(0.3, 0.5, int(puppy.sum()))
    ext[puppy, 0] = rng.uniform(0.2, 0.3, int(puppy.sum()))
    ext[puppy, 2] = rng.uniform(0.8, 1.6, int(puppy.sum())) # long in Z: direction 3
    sph = kind < sphere_frac
    ext[sph] = rng.uniform(0.2, 1.0, int(sph.sum()))[:, None]
    radius, length, yoffset = np.zeros(n), np.zeros(n), np.zeros(n)
    inertia = np.zeros((n, 3))
    mass = rng.uniform(0.5, 5.0, n)
    r, l, off, ro, d, I = C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_int(), (C.c_double * 3)()
    for i in range(n):
        L.clapgpu_capsule_geom(float(ext[i, 0]), float(ext[i, 1]), float(ext[i, 2]), 0.0, 0.0, C.byref(r), C.byref(l),
                               C.byref(off), C.byref(d), C.byref(ro))
        radius[i], length[i], yoffset[i] = r.value, l.value, off.value
        if l.value:
            L.clapgpu_mass_capsule_total(float(mass[i]), d.value, float(r.value), float(l.value), I)
        else:
            L.clapgpu_mass_sphere_total(float(mass[i]), float(r.value), I)
        inertia[i] = I[0], I[1], I[2]
    ang = rng.uniform(-math.pi, math.pi, (n, 3))
    q = quat_from_euler_xyz(ang[:, 0], ang[:, 1], ang[:, 2]).astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    lvel = rng.normal(0, 1, (n, 3))
    avel = rng.normal(0, 2.0, (n, 3))
    rest = rng.uniform(0, 1, n) < resting_frac
    lvel[rest] *= 1e-3
    avel[rest] *= 1e-3
    bflags = np.full(n, 2 | 8, np.uint32)                    # auto-disable (physics.c:1039) + gyroscopic (dBodyCreate)
    bflags[rest] |= 4
    return dict(n=n, pos=rng.uniform(0, box, (n, 3)), quat=q[:, [3, 0, 1, 2]].copy(), lvel=lvel, avel=avel, mass=mass,
                radius=radius, length=length, inertia=inertia, yoffset=yoffset, bflags=bflags,
                adis_steps_left=np.full(n, 30, np.int32), adis_time_left=np.zeros(n),
                body_entity=(entity_base + np.arange(n)).astype(np.int32),
                cell=float((length + 2 * radius).max()) if n else 1.0)


def static_boxes(n=64, box=64.0, seed=5):
    """Static collision geoms (the ground_space): AABBs as ODE stores them (minx,maxx,miny,maxy,minz,maxz)."""
    rng = _rng(seed)
    lo = rng.uniform(0, box, (n, 3))
    ext = rng.uniform(1, box / 4, (n, 3))
    out = np.empty((n, 6))
    out[:, 0::2] = lo
    out[:, 1::2] = lo + ext
    if n:
        out[0] = [-1e3, 1e3, -10.0, 0.5, -1e3, 1e3]          # a ground slab
    return out


LIGHTS_MAX = 128            # shader_constants.h:8
LIGHT_TILE = 64             # TILE_WIDTH, shader_constants.h:16 (light.c:210)


def lights(n=LIGHTS_MAX, seed=7, extent=60.0, n_dir=2, inactive_frac=0.1):
    """Light slots as light_get / light_set_* leave them (light.c:311-340, 473-520): a few
    directional slots (attenuation 1,0,0), point lights with glTF-style attenuation
    (1, linear, quadratic) scattered around the origin -- in front of, beside and behind a camera
    at the origin looking down -Z -- and some released slots inside [0, nr_lights)."""
    rng = _rng(seed)
    pos = np.stack([rng.uniform(-extent, extent, n), rng.uniform(-extent / 4, extent / 4, n),
                    rng.uniform(-2 * extent, extent / 2, n)], 1).astype(F32)
    color = (rng.uniform(0.2, 1.0, (n, 3)) * rng.uniform(0.5, 4.0, (n, 1))).astype(F32)
    att = np.stack([np.ones(n), rng.uniform(0.02, 0.8, n), np.exp(rng.uniform(math.log(0.02), math.log(40.0), n))],
                   1).astype(F32)
    is_dir = np.zeros(n, np.int32)
    is_dir[:min(n_dir, n)] = 1
    att[is_dir != 0] = (1, 0, 0)
    active = (rng.uniform(0, 1, n) >= inactive_frac).astype(np.uint32)
    if n:
        active[0] = 1
    return dict(nr_lights=n, pos=pos, color=color, attenuation=att, is_dir=is_dir, active=active)


POS_HISTORY_MAX = 8         # character.h:21


def character_feed(n=1000, seed=13, limbo_height=70.0, with_bodies=True, body_base=0, entity_base=0):
    """Per-character feeder state (character.c:546-611): position histories in every state of the
    ring (empty, partly filled, wrapped, a stored origin), entity positions near their last
    grounded spot or far below it (limbo), airborne flags.  limbo_height: scene.c default 70."""
    rng = _rng(seed)
    H = POS_HISTORY_MAX
    hist_pos = np.zeros((n, H, 3), F32)
    head = rng.integers(0, H, n).astype(np.uint32)
    wrapped = (rng.uniform(0, 1, n) < 0.4).astype(np.uint8)
    empty = rng.uniform(0, 1, n) < 0.15
    head[empty] = 0
    wrapped[empty] = 0
    ground = np.stack([rng.uniform(-200, 200, n), rng.uniform(0, 40, n), rng.uniform(-200, 200, n)], 1)
    for k in range(H):
        filled = (k < head) | (wrapped != 0)
        hist_pos[filled, k] = (ground + rng.normal(0, 2, (n, 3)))[filled]
    origin = rng.uniform(0, 1, n) < 0.05                     # newest entry exactly (0,0,0): never teleports
    newest = np.where(head > 0, head - 1, H - 1)
    hist_pos[origin, newest[origin]] = 0
    pos = ground + rng.normal(0, 1, (n, 3))
    fallen = rng.uniform(0, 1, n) < 0.3
    pos[fallen, 1] -= rng.uniform(0.5 * limbo_height, 3 * limbo_height, fallen.sum())
    edge = rng.uniform(0, 1, n) < 0.05                       # exactly limbo_height below the newest entry
    last = hist_pos[np.arange(n), newest]
    pos[edge, 1] = (last[edge, 1] - F32(limbo_height)).astype(F32)
    d = dict(n=n, entity=(entity_base + np.arange(n)).astype(np.uint32),
             body=((body_base + np.arange(n)) if with_bodies else np.full(n, -1)).astype(np.int32),
             hist_pos=hist_pos, hist_head=head, hist_wrapped=wrapped,
             airborne=(rng.uniform(0, 1, n) < 0.2).astype(np.uint8), pos=pos.astype(F32),
             limbo_height=float(limbo_height))
    return d
