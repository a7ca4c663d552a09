"""CPU: the physics restatement.  PARITY UNPINNED against the reference (ODE is an absent
submodule and the reference tests nothing here); what CAN be pinned is pinned: the phys_step
schedule literally from physics.c:773-787, and the restatement's internal consistency
(closed-form free fall, quaternion norm, brute-force broadphase, auto-disable bookkeeping)."""
import numpy as np

from clap_amd import synth
from oracle import binding as ob


def test_phys_step_schedule_matches_physics_c():
    """physics.c:773-787 hand-evaluated: 1/120 s substeps, at most 5, accumulator reset at 5."""
    fixed = 1.0 / 120.0
    acc = 0.0
    steps, acc = ob.phys_step_schedule(acc, 1.0 / 60.0)           # a 60 Hz frame -> 2 substeps
    assert steps == 2 and abs(acc) < 1e-12
    steps, acc = ob.phys_step_schedule(acc, 0.004)                 # less than a substep: accumulate
    assert steps == 0 and acc == 0.004
    steps, acc = ob.phys_step_schedule(acc, 0.005)                 # 0.009 -> 1 substep
    assert steps == 1 and abs(acc - (0.009 - fixed)) < 1e-15
    steps, acc = ob.phys_step_schedule(0.0, 1.0)                   # a hitch: clamp to 5 and drop the rest
    assert steps == 5 and acc == 0.0
    steps, acc = ob.phys_step_schedule(0.0, 5 * fixed)             # exactly 5: also resets
    assert steps == 5 and acc == 0.0


def test_world_defaults_are_the_reference_values():
    w = ob.world_defaults()
    assert tuple(w.gravity) == (0.0, -9.8, 0.0)                    # physics.c:1125
    assert w.linear_damping == 0.001                               # physics.c:1129
    assert (w.adis_linear_threshold_sq, w.adis_angular_threshold_sq, w.adis_steps) == (0.05 ** 2, 0.05 ** 2, 30)


def test_free_fall_and_rotation():
    b = synth.sphere_bodies(200, seed=1)
    st = ob.bodies_state(b)
    v0, p0, w0 = st["lvel"].copy(), st["pos"].copy(), st["avel"].copy()
    h, k = 1.0 / 120.0, 24
    for _ in range(k):
        ob.bodies_step(b, st, h)
    # semi-implicit Euler with damping 0.001 per step: bounded deviation from the undamped closed form
    v_exp = v0 + np.asarray([0, -9.8, 0]) * h * k
    assert np.allclose(st["lvel"], v_exp, rtol=0.03, atol=0.03)
    assert np.allclose(np.linalg.norm(st["quat"], axis=1), 1.0, atol=1e-14)
    assert np.array_equal(st["avel"], w0 + 0.0), "torque-free spheres keep their angular velocity"
    assert np.all(st["pos"][:, 1] < p0[:, 1] + np.abs(v0[:, 1]) * h * k + 1e-9)


def test_auto_disable_bookkeeping():
    b = synth.sphere_bodies(50, seed=2, resting_frac=1.0)
    st = ob.bodies_state(b)
    for step in range(31):
        st["bflags"] |= 16                                  # every body holds a (contact) joint: jointless bodies never sleep
        ob.bodies_step(b, st, 1.0 / 120.0)
        disabled = (st["bflags"] & 1) != 0
        assert disabled.all() == (step >= 29), f"step {step}: idle bodies sleep after 30 idle steps"
    assert not st["lvel"].any() and not st["avel"].any()
    b2 = synth.sphere_bodies(50, seed=3)
    st2 = ob.bodies_state(b2)
    ob.bodies_step(b2, st2, 1.0 / 120.0)
    assert (st2["adis_steps_left"] == 30).all() and not (st2["bflags"] & 1).any()


def test_broadphase_against_brute_force():
    rng = np.random.Generator(np.random.PCG64(9))
    n = 700
    pos = rng.uniform(0, 8, (n, 3))
    rad = rng.uniform(0.1, 0.5, n)
    pos[10] = pos[11] + [rad[10] + rad[11], 0, 0]                  # exactly touching on x: counts as overlap
    got = ob.broadphase_pairs(pos, rad)
    lo, hi = pos - rad[:, None], pos + rad[:, None]
    ov = np.all((lo[:, None, :] <= hi[None, :, :]) & (hi[:, None, :] >= lo[None, :, :]), axis=2)
    exp = np.argwhere(np.triu(ov, 1)).astype(np.uint32)
    assert np.array_equal(got, exp) and len(exp) > 100
    statics = synth.static_boxes(20, 8.0)
    sp = ob.broadphase_static_pairs(statics, pos, rad)
    slo, shi = statics[:, 0::2], statics[:, 1::2]
    ov2 = np.all((lo[:, None, :] <= shi[None, :, :]) & (hi[:, None, :] >= slo[None, :, :]), axis=2)
    assert np.array_equal(sp, np.argwhere(ov2).astype(np.uint32))


def test_body_readback_layout():
    b = synth.sphere_bodies(40, seed=4, entity_base=5)
    st = ob.bodies_state(b)
    ps = np.zeros((50, 4), np.float32); ps[:, 3] = 1
    rot = np.zeros((50, 4), np.float32)
    fl = np.zeros(50, np.uint32)
    moving = ob.phys_body_update(b, st, ps, rot, fl)
    assert np.array_equal(ps[5:45, 0], st["pos"][:, 0].astype(np.float32))
    assert np.array_equal(ps[5:45, 1], (st["pos"][:, 1] - b["yoffset"]).astype(np.float32))   # physics.c:799
    assert np.array_equal(rot[5:45], st["quat"][:, [1, 2, 3, 0]].astype(np.float32))          # wxyz -> xyzw
    assert (fl[5:45] == synth.E_DIRTY).all() and not fl[:5].any()
    assert np.array_equal(moving, (np.linalg.norm(st["lvel"], axis=1) > 1e-3).astype(np.uint8))


def test_sphere_contact_restatement_properties():
    """dCollideSpheres + phys_contact_surface restated (parity unpinned: ODE absent): invariants."""
    b = synth.sphere_bodies(4000, box=12.0, seed=8)
    pairs = ob.broadphase_pairs(b["pos"], b["radius"], max_pairs=1 << 20)
    c, total = ob.contacts_spheres(pairs, b["pos"], b["radius"])
    hit = c["nc"] == 1
    assert total == hit.sum() and 0 < total < len(pairs)
    d = np.linalg.norm(b["pos"][pairs[:, 0]] - b["pos"][pairs[:, 1]], axis=1)
    rsum = b["radius"][pairs[:, 0]] + b["radius"][pairs[:, 1]]
    assert np.array_equal(hit, d <= rsum)
    assert np.allclose(np.linalg.norm(c["normal"][hit], axis=1), 1.0, atol=1e-12)
    assert np.allclose(c["depth"][hit], (rsum - d)[hit], atol=1e-12) and np.all(c["depth"][hit] >= 0)
    # the contact point lies on the centre line, half the penetration inside sphere 1's surface
    p1 = b["pos"][pairs[:, 0]][hit]
    on_surface = p1 - c["normal"][hit] * b["radius"][pairs[:, 0]][hit, None]
    assert np.allclose(c["pos"][hit], on_surface + c["normal"][hit] * (c["depth"][hit, None] / 2), atol=1e-9)
    assert np.all(c["mode"][hit] == 0x18) and np.all(c["soft_erp"][hit] == 0.05) and np.all(c["soft_cfm"][hit] == 0.01)
    assert not c[~hit].tobytes().strip(b"\0"), "non-touching pairs leave a zero record"
    mat = np.tile(np.asarray([0.5, 0.1, 0.9, 0.0, 0.0]), (4000, 1))
    mat[::2] = [0.2, 0.3, 0.4, 0.1, 0.02]
    c2, _ = ob.contacts_spheres(pairs, b["pos"], b["radius"], mat)
    k = np.flatnonzero(hit & (pairs[:, 0] % 2 == 0) & (pairs[:, 1] % 2 == 1))[0]
    assert c2["bounce"][k] == 0.5 and c2["bounce_vel"][k] == 0.2 and c2["mu"][k] == np.sqrt(0.4 * 0.9)
    assert c2["soft_erp"][k] == 0.1 and c2["soft_cfm"][k] == 0.02 and c2["mode"][k] == 0x1c


def test_sphere_box_contact_restatement_properties():
    """dCollideSphereBox for axis-aligned boxes + phys_contact_surface restated (parity unpinned: ODE absent):
    invariants against an independent numpy formulation (closest point on the box)."""
    rng = np.random.Generator(np.random.PCG64(12))
    n, ns = 6000, 40
    lo = rng.uniform(-10, 8, (ns, 3))
    hi = lo + rng.uniform(0.5, 6, (ns, 3))
    aabb = np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    pos = rng.uniform(-12, 12, (n, 3))
    radius = rng.uniform(0.1, 1.5, n)
    pairs = np.stack([rng.integers(0, n, 20000), rng.integers(0, ns, 20000)], 1).astype(np.uint32)
    # special cases: a centre exactly on a face, exactly on a corner, exactly in the middle of a cube, grazing contact
    pos[0] = [hi[0, 0], (lo[0, 1] + hi[0, 1]) / 2, (lo[0, 2] + hi[0, 2]) / 2]
    pos[1] = hi[1]
    lo[2] = [0, 0, 0]; hi[2] = [2, 2, 2]; aabb[2] = [0, 2, 0, 2, 0, 2]; pos[2] = [1, 1, 1]
    pos[3] = [hi[3, 0] + radius[3], (lo[3, 1] + hi[3, 1]) / 2, (lo[3, 2] + hi[3, 2]) / 2]
    pairs[:4] = [[0, 0], [1, 1], [2, 2], [3, 3]]
    c, total = ob.contacts_sphere_box(pairs, pos, radius, aabb)
    hit = c["nc"] == 1
    P, R = pos[pairs[:, 0]], radius[pairs[:, 0]]
    LO, HI = lo[pairs[:, 1]], hi[pairs[:, 1]]
    closest = np.clip(P, LO, HI)
    dist = np.linalg.norm(P - closest, axis=1)
    inside = np.all((P >= LO) & (P <= HI), axis=1)
    assert total == hit.sum() and 0 < total < len(pairs)
    far = dist > R + 1e-9
    near = dist < R - 1e-9
    assert not hit[far & ~inside].any() and hit[near | inside].all()
    out = hit & ~inside
    assert np.allclose(c["depth"][out], (R - dist)[out], atol=1e-12) and np.all(c["depth"][hit] >= 0)
    assert np.allclose(c["pos"][out], closest[out], atol=1e-12)
    nz = out & (dist > 1e-9)
    assert np.allclose(c["normal"][nz], ((P - closest) / np.maximum(dist, 1e-300)[:, None])[nz], atol=1e-12)
    assert np.allclose(np.linalg.norm(c["normal"][hit], axis=1), 1.0, atol=1e-12)
    ins = hit & inside
    face = np.minimum(P - LO, HI - P)[ins]                  # distance to the faces, per axis
    assert np.allclose(c["depth"][ins], face.min(axis=1) + R[ins], atol=1e-12)
    assert np.all(np.abs(c["normal"][ins]).sum(axis=1) == 1.0) and np.array_equal(c["pos"][ins], P[ins])
    # the named cases
    assert c["nc"][0] == 1 and c["depth"][0] == radius[0] and tuple(c["normal"][0]) == (1.0, 0.0, 0.0)   # on a face: inside branch
    assert c["nc"][1] == 1 and c["depth"][1] == radius[1]
    assert c["nc"][2] == 1 and tuple(c["normal"][2]) == (-1.0, 0.0, 0.0) and c["depth"][2] == 1.0 + radius[2]   # ties: first axis, t = 0 -> -1
    assert c["nc"][3] == 1 and abs(c["depth"][3]) < 1e-12
    assert np.all(c["mode"][hit] == 0x18) and np.all(c["soft_erp"][hit] == 0.05) and np.all(c["soft_cfm"][hit] == 0.01)
    assert not c[~hit].tobytes().strip(b"\0"), "non-touching pairs leave a zero record"
    mat = np.tile(np.asarray([0.5, 0.1, 0.9, 0.0, 0.0]), (n, 1))
    smat = np.tile(np.asarray([0.2, 0.3, 0.4, 0.1, 0.02]), (ns, 1))
    c2, _ = ob.contacts_sphere_box(pairs, pos, radius, aabb, mat, smat)
    k = int(np.flatnonzero(hit)[0])
    assert c2["bounce"][k] == 0.5 and c2["bounce_vel"][k] == 0.2 and c2["mu"][k] == np.sqrt(0.9 * 0.4)
    assert c2["soft_erp"][k] == 0.1 and c2["soft_cfm"][k] == 0.02 and c2["mode"][k] == 0x1c
    c3, _ = ob.contacts_sphere_box(pairs, pos, radius, aabb, mat, None)      # one side without parameters: defaults
    assert c3["mu"][k] == 0.0 and c3["mode"][k] == 0x18
