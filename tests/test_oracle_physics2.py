"""CPU: the round-2 physics restatement (oracle/physics2.c) -- PARITY UNPINNED against the reference (ODE is an
absent submodule), so these tests hold it to what can be checked without ODE: closed forms (capsule inertia,
the offset rotation), conservation laws (a torque-free body keeps |L| under the implicit gyroscopic step),
geometric invariants of every collider against independent formulations (sampled closest points), and the
product's host helpers computing the same numbers."""
import ctypes as C
import math

import numpy as np
import pytest

from clap_amd import _lib, synth
from oracle import binding as ob


def test_offset_rotation_maps_capsule_axis_to_y():
    R = ob.geom_offset_rotation().reshape(3, 4)[:, :3]
    assert np.allclose(R @ [0, 0, 1], [0, 1, 0], atol=1e-15)          # ODE capsules run along local Z; CLAP wants Y
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)
    Rg = (C.c_double * 12)()
    _lib.lib().clapgpu_geom_offset_rotation(Rg)
    assert np.array_equal(np.array(Rg), ob.geom_offset_rotation())


def test_capsule_mass_matches_numerical_integration():
    a, b_, total = 0.3, 1.1, 2.5
    I = ob.mass_capsule_total(total, 3, a, b_)
    rng = np.random.Generator(np.random.PCG64(1))
    p = rng.uniform([-a, -a, -b_ / 2 - a], [a, a, b_ / 2 + a], (2_000_000, 3))
    zc = np.clip(p[:, 2], -b_ / 2, b_ / 2)
    inside = p[:, 0] ** 2 + p[:, 1] ** 2 + (p[:, 2] - zc) ** 2 <= a * a
    q = p[inside]
    m = total / len(q)
    num = [m * (q[:, 1] ** 2 + q[:, 2] ** 2).sum(), m * (q[:, 0] ** 2 + q[:, 2] ** 2).sum(), m * (q[:, 0] ** 2 + q[:, 1] ** 2).sum()]
    assert np.allclose(I, num, rtol=5e-3)
    Is = ob.mass_sphere_total(total, a)
    assert np.allclose(Is, 0.4 * total * a * a)
    Ig = (C.c_double * 3)()
    _lib.lib().clapgpu_mass_capsule_total(total, 3, a, b_, Ig)
    assert np.array_equal(np.array(Ig), I)
    _lib.lib().clapgpu_mass_sphere_total(total, a, Ig)
    assert np.array_equal(np.array(Ig), Is)


def test_capsule_geom_follows_phys_geom_capsule_new():
    # upright: Y largest -> direction 2, r = min / 2, length = Y / 2 - 2 r, yoffset = Y / 2
    r, l, off, d, ro = ob.capsule_geom(0.4, 1.8, 0.5)
    assert (d, r) == (2, np.float32(0.2)) and l == np.float32(np.float32(0.9) - np.float32(0.4)) and off == np.float32(0.9)
    assert ro == np.float32(r + l / 2)
    # puppy: Z largest -> direction 3
    r, l, off, d, ro = ob.capsule_geom(0.3, 0.5, 1.4)
    assert d == 3 and r == np.float32(0.15) and l == np.float32(np.float32(1.4) - np.float32(0.3)) and ro == r
    # a cube: Y / 2 - 2 r = 0 -> a sphere
    r, l, off, d, ro = ob.capsule_geom(1.0, 1.0, 1.0)
    assert l == 0.0 and r == 0.5
    # X largest: direction stays 1 for the mass, geometry as upright (the reference's fall-through)
    assert ob.capsule_geom(2.0, 1.0, 0.5)[3] == 1
    rng = np.random.Generator(np.random.PCG64(3))
    L = _lib.lib()
    for X, Y, Z, gr, go in rng.uniform(0.1, 2.0, (500, 5)) * [1, 1, 1, 0.3, 1] * (rng.uniform(0, 1, (500, 5)) > [0, 0, 0, .5, .5]):
        out = [C.c_float(), C.c_float(), C.c_float(), C.c_int(), C.c_float()]
        L.clapgpu_capsule_geom(X, Y, Z, gr, go, C.byref(out[0]), C.byref(out[1]), C.byref(out[2]), C.byref(out[3]), C.byref(out[4]))
        assert tuple(o.value for o in out) == ob.capsule_geom(X, Y, Z, gr, go)


def test_gyroscopic_step_conserves_angular_momentum_magnitude():
    b = synth.capsule_bodies(2000, box=20.0, seed=7)
    b["bflags"] = np.full(2000, 8 | 4, np.uint32)                        # gyroscopic, no gravity
    st = ob.bodies_state(b)

    def momentum(q, om):
        w, x, y, z = q.T
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                      np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                      np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)
        body = np.einsum("nji,nj->ni", R, om)
        return np.einsum("nij,nj->ni", R, body * b["inertia"])
    L0 = momentum(st["quat"], st["avel"])
    for _ in range(240):                                                 # two seconds
        ob.bodies_step(b, st, 1 / 120)
    L1 = momentum(st["quat"], st["avel"])
    caps = b["length"] > 0
    assert not np.allclose(st["avel"][caps], b["avel"][caps]), "anisotropic bodies precess"
    drift = np.linalg.norm(L1 - L0, axis=1) / np.linalg.norm(L0, axis=1)
    assert np.median(drift) < 0.02 and drift.max() < 0.2, (np.median(drift), drift.max())
    sph = ~caps                                                          # isotropic inertia: the spin does not change
    assert np.allclose(st["avel"][sph], b["avel"][sph], rtol=1e-12, atol=1e-13)
    assert np.allclose(np.linalg.norm(st["quat"], axis=1), 1.0, atol=1e-14)


def test_jointless_bodies_never_sleep_and_window_mean_decides():
    b = synth.capsule_bodies(50, box=5.0, seed=1, resting_frac=1.0)
    st = ob.bodies_state(b)
    for _ in range(40):
        ob.bodies_step(b, st, 1 / 120)
    assert not (st["bflags"] & 1).any(), "no joint, no sleep (ODE: don't freeze objects mid-air)"
    for _ in range(31):
        st["bflags"] |= 16
        ob.bodies_step(b, st, 1 / 120)
    assert (st["bflags"] & 1).all()
    assert not st["lvel"].any() and not st["avel"].any()


def _seg_dist(p1, q1, p2, q2, samples=400):
    t = np.linspace(0, 1, samples)
    a = p1[None] + (q1 - p1)[None] * t[:, None]
    c = p2[None] + (q2 - p2)[None] * t[:, None]
    return np.sqrt(((a[:, None] - c[None]) ** 2).sum(-1)).min()


def test_capsule_capsule_and_capsule_sphere_against_sampled_distances():
    rng = np.random.Generator(np.random.PCG64(5))
    n = 400
    pos = rng.uniform(0, 3.0, (n, 3))
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    radius = rng.uniform(0.1, 0.4, n)
    length = np.where(rng.uniform(0, 1, n) < 0.3, 0.0, rng.uniform(0.2, 1.5, n))
    G = ob.geoms(n, pos=pos, axis=ax, radius=radius, length=length)
    pairs = np.stack(np.triu_indices(n, 1), 1).astype(np.uint32)[::37]
    rec, tot = ob.contacts_geoms(pairs, G, G)
    checked = 0
    for (i, j), c in zip(pairs, rec):
        d = _seg_dist(pos[i] + ax[i] * length[i] / 2, pos[i] - ax[i] * length[i] / 2,
                      pos[j] + ax[j] * length[j] / 2, pos[j] - ax[j] * length[j] / 2)
        gap = d - radius[i] - radius[j]
        if abs(gap) < 2e-3:
            continue                                                     # sampling resolution
        assert (c["nc"] > 0) == (gap < 0), (i, j, gap, c["nc"])
        if c["nc"] == 1 and abs(1 - abs(ax[i] @ ax[j])) > 1e-3:
            assert abs(c["depth"] - (-gap)) < 5e-3
            assert abs(np.linalg.norm(c["normal"]) - 1) < 1e-12
            checked += 1
    assert checked > 50 and tot > 0


def test_parallel_capsules_two_contacts_and_reversal_symmetry():
    pos = np.array([[0, 0, 0], [0.3, 0.2, 0.0], [5, 5, 5]], float)
    ax = np.array([[0, 1, 0], [0, 1, 0], [1, 0, 0]], float)
    G = ob.geoms(3, pos=pos, axis=ax, radius=np.array([0.2, 0.2, 0.3]), length=np.array([1.0, 0.6, 0.0]))
    rec, tot = ob.contacts_geoms(np.array([[0, 1]], np.uint32), G, G)
    c = rec[0]
    assert c["nc"] == 2 and tot == 1
    assert np.allclose(c["normal"], [-1, 0, 0]) and np.allclose(c["normal2"], [-1, 0, 0])
    assert np.isclose(c["depth"], 0.1) and np.isclose(c["depth2"], 0.1)
    assert np.isclose(c["pos"][1], -0.1) and np.isclose(c["pos2"][1], 0.5), "the ends of the overlap interval"
    # sphere (g1) vs capsule (g2) = capsule-sphere swapped, normal negated
    pos2 = np.array([[0.35, 0.1, 0.0], [0, 0, 0]], float)
    H = ob.geoms(2, pos=pos2, axis=np.array([[0, 0, 1], [0, 1, 0]], float), radius=np.array([0.2, 0.2]), length=np.array([0.0, 1.0]))
    fwd, _ = ob.contacts_geoms(np.array([[1, 0]], np.uint32), H, H)
    rev, _ = ob.contacts_geoms(np.array([[0, 1]], np.uint32), H, H)
    assert fwd["nc"][0] == rev["nc"][0] == 1 and fwd["depth"][0] == rev["depth"][0]
    assert np.array_equal(fwd["normal"][0], -rev["normal"][0]) and np.array_equal(fwd["pos"][0], rev["pos"][0])


def test_capsule_box_against_sampled_distance_and_deep_flag():
    rng = np.random.Generator(np.random.PCG64(8))
    box = np.array([[-1.0, 1.0, -0.5, 0.5, -2.0, 2.0]])
    n = 600
    pos = rng.uniform(-3, 3, (n, 3))
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    radius, length = rng.uniform(0.1, 0.5, n), rng.uniform(0.2, 2.0, n)
    A = ob.geoms(n, pos=pos, axis=ax, radius=radius, length=length)
    B = ob.geoms(1, kind=np.array([2], np.uint8), aabb=box)
    pairs = np.stack([np.arange(n), np.zeros(n)], 1).astype(np.uint32)
    rec, _ = ob.contacts_geoms(pairs, A, B)
    t = np.linspace(-0.5, 0.5, 2001)
    deep = touching = apart = 0
    for i, c in enumerate(rec):
        pts = pos[i][None] + ax[i][None] * (t * length[i])[:, None]
        clamped = np.clip(pts, box[0, 0::2], box[0, 1::2])
        d = np.sqrt(((pts - clamped) ** 2).sum(1)).min()
        if d < 1e-3:
            assert c["nc"] in (0x80000000, 1), "axis through the box: ODE's dBoxBox case (or a grazing one)"
            deep += c["nc"] == 0x80000000
            continue
        gap = d - radius[i]
        if abs(gap) < 2e-3:
            continue
        assert (c["nc"] == 1) == (gap < 0), (i, gap, c["nc"])
        if c["nc"] == 1:
            assert abs(c["depth"] + gap) < 5e-3 and abs(np.linalg.norm(c["normal"]) - 1) < 1e-12
            touching += 1
        else:
            apart += 1
    assert deep > 20 and touching > 50 and apart > 50


def test_sweep_stops_at_the_floor():
    # an upright capsule (half length 0.25 + radius 0.25) whose lower cap ends 1.25 above a slab, moving 3.0 down:
    # it travels 1.25 and reports the slab's up normal
    pos = np.array([[0.0, 0.5 + 0.25 + 1.0, 0.0]])
    A = ob.geoms(1, pos=pos, axis=np.array([[0.0, 1.0, 0.0]]), radius=np.array([0.25]), length=np.array([0.5]))
    S = ob.geoms(1, kind=np.array([2], np.uint8), aabb=np.array([[-10, 10, -1.0, 0.0, -10, 10]], float))
    frac, normal, hit = ob.sweep_capsule(A, 0, np.array([0, -3.0, 0], np.float32), S, np.array([0], np.uint32))
    assert hit == -2 and np.allclose(normal, [0, 1, 0])
    assert abs(frac * 3.0 - 1.25) < 0.02
    frac, normal, hit = ob.sweep_capsule(A, 0, np.array([0, 0.5, 0], np.float32), S, np.array([0], np.uint32))
    assert frac == 1.0 and hit == -1 and tuple(normal) == (0, 1, 0)
    frac, _n, _h = ob.sweep_capsule(A, 0, np.zeros(3, np.float32), S, np.array([0], np.uint32))
    assert frac == 1.0
