"""GPU: rigid-body integrate, read-back and broadphase (through the C ABI) against the oracle.

The oracle for this block is parity-UNPINNED against the reference (ODE absent), so these tests
establish GPU == restatement: pair sets bit-exact (fp64 comparisons), body state after N steps
within 1e-5 relative (in practice bit-exact: both sides are IEEE fp64 without contraction)."""
import numpy as np
import pytest

from clap_amd import synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-300)


def oracle_aabbs(b):
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    return st["aabb"]


def check_broadphase(world, b, statics, cap, aabb=None):
    out = world.download()
    bb = oracle_aabbs(b) if aabb is None else aabb              # aabb: the oracle's boxes after its own steps
    assert np.array_equal(out["aabb"].view(np.uint64), bb.view(np.uint64)), "geom AABBs"
    exp = ob.broadphase_aabb_pairs(bb, max_pairs=cap)
    assert out["pair_total"] == len(exp)
    assert np.array_equal(out["pairs"], exp), "body x body candidate pairs (ascending list)"
    if statics is not None:
        exp_s = ob.broadphase_aabb_static_pairs(statics, bb, max_pairs=cap)
        assert out["static_pair_total"] == len(exp_s)
        assert np.array_equal(out["static_pairs"], exp_s), "body x static candidate pairs (ascending list)"
    assert world.broadphase_status() == 0
    return exp


@pytest.mark.parametrize("n,box", [(1, 4.0), (63, 4.0), (5000, 16.0), (40_000, 32.0)], ids=["n1", "n63", "n5k", "n40k"])
@pytest.mark.parametrize("kind", ["spheres", "capsules"])
def test_broadphase_pair_set_exact(n, box, kind, cuda_device):
    from clap_amd import physics
    b = synth.sphere_bodies(n, box=box, seed=21) if kind == "spheres" else synth.capsule_bodies(n, box=box * 1.6, seed=21)
    statics = synth.static_boxes(37, box)
    cap = max(64 * n, 1024)
    world = physics.PhysWorld(b, statics, pair_capacity=cap, device=cuda_device)
    for _ in range(2):                                      # twice: the counters the kernels leave behind are clean
        world.broadphase()
    exp = check_broadphase(world, b, statics, cap)
    if n >= 5000:
        assert len(exp) > n // 10


def test_broadphase_thousands_of_statics_small_and_large(cuda_device):
    """The ground_space as the reference fills it: thousands of small static colliders (binned per block), a few
    big ones (terrain-sized AABBs, tested by every block), degenerate ones (zero extent, inverted = never)."""
    from clap_amd import physics
    n, box = 30_000, 40.0
    b = synth.capsule_bodies(n, box=box, seed=5)
    rng = np.random.Generator(np.random.PCG64(9))
    ns = 6000
    lo = rng.uniform(-2, box, (ns, 3))
    ext = rng.uniform(0.05, 3.0, (ns, 3))
    statics = np.empty((ns, 6))
    statics[:, 0::2], statics[:, 1::2] = lo, lo + ext
    statics[0] = [-1e4, 1e4, -50, 0.7, -1e4, 1e4]          # terrain slab
    statics[1] = [5, 35, 0, 40, 5, 6]                      # a long wall: many blocks
    statics[2] = [10, 10, 10, 10, 10, 10]                  # a point
    statics[3] = [12, 11, 12, 11, 12, 11]                  # inverted: overlaps nothing
    statics[4] = [-1e300, 1e300, -1e300, 1e300, -1e300, 1e300]
    cap = 2_000_000
    world = physics.PhysWorld(b, statics, pair_capacity=cap, device=cuda_device)
    world.broadphase()
    check_broadphase(world, b, statics, cap)
    out = world.download()
    assert out["static_pair_total"] > n, "every body touches the all-space box"


def test_broadphase_touching_and_negative_coordinates(cuda_device):
    """Touching AABBs collide (ODE's separation test is strict); cells left of the origin hash fine."""
    from clap_amd import physics
    b = synth.sphere_bodies(400, box=6.0, seed=5)
    b["pos"] -= 3.0
    b["pos"][7] = b["pos"][8] + [b["radius"][7] + b["radius"][8], 0, 0]
    b["pos"][20] = b["pos"][21]
    world = physics.PhysWorld(b, None, pair_capacity=100_000, device=cuda_device)
    world.broadphase()
    exp = check_broadphase(world, b, None, 100_000)
    assert any((p == [7, 8]).all() for p in exp) and any((p == [20, 21]).all() for p in exp)


def test_broadphase_dense_bodies_with_long_partner_lists(cuda_device):
    """Crowded cells: most bodies have more partners than a body's fixed list slot holds, so their lists go
    through the arena and the emit pass ranks them there."""
    from clap_amd import physics
    b = synth.sphere_bodies(3000, box=6.0, seed=6)
    exp = ob.broadphase_aabb_pairs(oracle_aabbs(b), max_pairs=1 << 22)
    per_body = np.bincount(exp[:, 0], minlength=3000)
    assert per_body.max() > 16 and (per_body <= 16).any()
    world = physics.PhysWorld(b, None, pair_capacity=len(exp) + 8, device=cuda_device)
    world.broadphase()
    check_broadphase(world, b, None, 1 << 22)


def test_static_pairs_with_more_hits_than_the_kept_list(cuda_device):
    """Bodies inside 40 nested static boxes (more hits than the per-body list slot) next to bodies that hit few."""
    from clap_amd import physics
    b = synth.sphere_bodies(700, box=8.0, seed=9)
    statics = synth.static_boxes(60, 8.0, seed=2)
    for s_ in range(40):                                    # nested boxes around the low corner
        statics[s_] = [-1.0, 3.0 + 0.05 * s_, -1.0, 3.0 + 0.05 * s_, -1.0, 3.0 + 0.05 * s_]
    exp_s = ob.broadphase_aabb_static_pairs(statics, oracle_aabbs(b), max_pairs=1 << 20)
    per_body = np.bincount(exp_s[:, 0], minlength=700)
    assert per_body.max() > 16 and ((per_body > 0) & (per_body <= 16)).any()
    world = physics.PhysWorld(b, statics, pair_capacity=1 << 16, static_pair_capacity=len(exp_s) + 8, device=cuda_device)
    world.broadphase()
    check_broadphase(world, b, statics, 1 << 20)


def test_pair_capacity_overflow_reports_total(cuda_device):
    from clap_amd import physics
    b = synth.sphere_bodies(3000, box=6.0, seed=6)
    exp = ob.broadphase_aabb_pairs(oracle_aabbs(b), max_pairs=1 << 22)
    world = physics.PhysWorld(b, None, pair_capacity=100, device=cuda_device)
    world.broadphase()
    out = world.download()
    assert out["pair_total"] == len(exp) > 100          # total found is reported, only `capacity` written


def test_broadphase_flags_a_body_larger_than_the_cell(cuda_device):
    from clap_amd import physics
    b = synth.sphere_bodies(500, box=8.0, seed=2)
    b["radius"] = b["radius"].copy()
    b["radius"][17] = 3.0                                   # AABB edge 6 > cell 1
    world = physics.PhysWorld(b, None, device=cuda_device)
    world.broadphase()
    assert world.broadphase_status() & 1


def _run_steps(b, world, st, dts, joint_mask=None):
    import torch
    acc, total = world.time_acc.value, 0               # the accumulator carries over between calls
    for dt in dts:
        steps, acc = ob.phys_step_schedule(acc, dt)
        got = world.phys_step_begin(dt)
        assert got == steps and world.time_acc.value == acc
        for _ in range(steps):
            if joint_mask is not None:                       # what the contact pass does for touching bodies
                st["bflags"][joint_mask] |= 16
                world.bflags[torch.from_numpy(np.flatnonzero(joint_mask)).to(world.device)] |= 16
            ob.bodies_step(b, st, 1.0 / 120.0)
            world.world_step(1.0 / 120.0)
        total += steps
    return total


def _assert_state_equal(out, st):
    for k in ("pos", "quat", "lvel", "avel", "aabb", "axis"):
        assert rel_err(out[k], st[k]) <= 1e-5, k
        assert np.array_equal(out[k].view(np.uint64), st[k].view(np.uint64)), f"{k}: fp64 IEEE on both sides -> bit-exact"
    assert np.array_equal(out["bflags"], st["bflags"])
    assert np.array_equal(out["adis_steps_left"], st["adis_steps_left"])


@pytest.mark.parametrize("kind", ["spheres", "capsules"])
def test_integrate_and_schedule_match_oracle(kind, cuda_device):
    """dWorldQuickStep for bodies without constraint rows: schedule, gravity, the implicit gyroscopic torque of the
    capsules' anisotropic inertia, pose update, damping, the moved geoms' axis + AABB; then auto-disable of the
    resting bodies that hold a joint (jointless bodies never sleep, like ODE)."""
    from clap_amd import physics
    n = 20_000
    b = (synth.sphere_bodies(n, box=32.0, seed=8, resting_frac=0.2) if kind == "spheres"
         else synth.capsule_bodies(n, box=32.0, seed=8, resting_frac=0.2))
    world = physics.PhysWorld(b, None, device=cuda_device)
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    total = _run_steps(b, world, st, (1 / 60, 0.004, 0.005, 1 / 30, 0.3, 1 / 144))   # incl. a hitch that clamps to 5 substeps
    # physics.c:773-787 with fixed_dt = 1/120: 1/60 -> 2 substeps; 0.004 -> 0; + 0.005 = 0.009 -> 1; + 1/30 -> 4;
    # 0.3 s (a hitch) -> the 5-substep cap, accumulator reset; 1/144 -> 0
    assert total == 2 + 0 + 1 + 4 + 5 + 0
    _assert_state_equal(world.download(), st)
    if kind == "capsules":
        assert not np.array_equal(st["avel"], b["avel"]), "anisotropic inertia: the spin precesses"
    resting = (b["bflags"] & 4) != 0
    with_joint = resting & (np.arange(n) % 2 == 0)          # half of the resting bodies touch something
    _run_steps(b, world, st, [1 / 120] * 31, joint_mask=with_joint)
    out = world.download()
    _assert_state_equal(out, st)
    asleep = (out["bflags"] & 1) != 0
    assert asleep[with_joint].all() and not asleep[~with_joint].any(), "only resting bodies that hold a joint fall asleep"


def test_auto_disable_sample_window(cuda_device):
    """dBodySetAutoDisableAverageSamplesCount(b, 4): the idle test runs on the mean of the last four samples and
    only once the ring is full."""
    from clap_amd import physics
    n = 4000
    b = synth.capsule_bodies(n, box=20.0, seed=12, resting_frac=0.5)
    b["adis_average_samples"] = 4
    world = physics.PhysWorld(b, None, device=cuda_device)
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    everyone = np.ones(n, bool)
    _run_steps(b, world, st, [1 / 120] * 20, joint_mask=everyone)
    out = world.download()
    _assert_state_equal(out, st)
    assert np.array_equal(world.adis_counter.cpu().numpy().view(np.uint32), st["adis_counter"])
    assert np.array_equal(world.adis_samples.cpu().numpy(), st["adis_samples"])
    assert (out["adis_steps_left"] < 30).any() and (out["adis_steps_left"] == 30).any()
    _run_steps(b, world, st, [1 / 120] * 20, joint_mask=everyone)
    out = world.download()
    _assert_state_equal(out, st)
    assert (out["bflags"] & 1).any() and not (out["bflags"] & 1).all()


def test_body_readback_feeds_entity_update(cuda_device):
    """phys_step -> phys_body_update -> mq_update, as clap_frame orders them (clap.c:604,614)."""
    from clap_amd import entities, physics
    n = 3000
    scene = synth.pad_levels(synth.entities_flat(n, seed=3))
    b = synth.sphere_bodies(n, box=40.0, seed=3)
    batch = entities.EntityBatch(scene, cuda_device)
    world = physics.PhysWorld(b, None, device=cuda_device)
    fr, _v, _p = entities.view_calc_frustum(synth.camera(pos=(20, 20, 90)))
    fr_o, _vo, _po = ob.frustum_from_camera(synth.camera(pos=(20, 20, 90)))
    st_b = ob.bodies_state(b)
    st_e = ob.entity_state(scene)
    for frame in range(3):
        steps, _ = ob.phys_step_schedule(0.0, 1 / 60)
        for _ in range(steps):
            ob.bodies_step(b, st_b, 1 / 120)
        moving = ob.phys_body_update(b, st_b, scene["pos_scale"], scene["rot"], st_e["flags"])
        ob.entities_update(scene, st_e)
        vis, mask = ob.entities_cull(scene["n"], st_e["flags"], st_e["aabb"], fr_o)

        world.time_acc.value = 0.0
        world.phys_step(1 / 60, broadphase=False)
        world.phys_body_update(batch)
        batch.mq_update(fr)
        batch.compact_visible()
    out = batch.download()
    assert np.array_equal(out["mx"].view(np.uint32), st_e["mx"].view(np.uint32))
    assert np.array_equal(out["visible"], vis)
    assert moving.any()


@pytest.mark.parametrize("kind", ["spheres", "capsule_mix"])
def test_c4_body_count_full_size(kind, cuda_device):
    """BASELINE config 4 (body half) at full size: 262 144 bodies -- spheres, and the reference's own geoms (capsules,
    "puppy" capsules, spheres) -- against 64 large + 2 000 small statics: both pair lists equal the oracle's sweep-and-
    prune, strictly ascending, every pair overlapping."""
    from clap_amd import physics
    n = 262_144
    b = synth.sphere_bodies(n, box=64.0, seed=4) if kind == "spheres" else synth.capsule_bodies(n, box=60.0, seed=4)
    rng = np.random.Generator(np.random.PCG64(3))
    box = 64.0 if kind == "spheres" else 60.0
    small = np.empty((2000, 6))
    lo = rng.uniform(0, box, (2000, 3))
    small[:, 0::2], small[:, 1::2] = lo, lo + rng.uniform(0.1, 2.0, (2000, 3))
    statics = np.concatenate([synth.static_boxes(64, box), small])
    world = physics.PhysWorld(b, statics, pair_capacity=4_000_000, static_pair_capacity=8_000_000, device=cuda_device)
    world.broadphase()
    exp = check_broadphase(world, b, statics, 8_000_000)
    out = world.download()
    p = out["pairs"].astype(np.int64)
    assert (p[:, 0] < p[:, 1]).all()
    key = p[:, 0] * (1 << 32) + p[:, 1]
    assert (np.diff(key) > 0).all()
    bb = out["aabb"]
    assert np.all((bb[p[:, 0], 0::2] <= bb[p[:, 1], 1::2]) & (bb[p[:, 0], 1::2] >= bb[p[:, 1], 0::2]))
    assert 0.3 < len(exp) / b["n"] < 3.0
    # ... and the body half of __phys_step at the same size: schedule (incl. the 5-substep cap) + k_bodies_step (pose,
    # velocities, the moved geoms' axis and AABB) against oracle/physics2.c, fp64 on both sides: bit-exact; then the
    # broadphase once more over the MOVED boxes
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    total = _run_steps(b, world, st, (1 / 60, 0.3, 1 / 120))
    assert total == 2 + 5 + 1
    _assert_state_equal(world.download(), st)
    world.broadphase()
    check_broadphase(world, b, statics, 8_000_000, aabb=st["aabb"])


def test_sphere_contacts_match_oracle(cuda_device):
    """near_callback for spheres (8f rank 3): contact geometry and surface parameters per candidate
    pair, bit-exact against the restatement (IEEE fp64 on both sides); the count of touching pairs."""
    from clap_amd import physics
    n = 20_000
    b = synth.sphere_bodies(n, box=24.0, seed=33)
    b["pos"][11] = b["pos"][10]                             # coincident centres: normal (1,0,0), depth r1 + r2
    b["pos"][13] = b["pos"][12] + [b["radius"][12] + b["radius"][13], 0, 0]      # exactly touching
    rng = np.random.Generator(np.random.PCG64(4))
    mat = np.stack([rng.choice([0.0, 0.3, 0.8], n), rng.uniform(0, 0.2, n), rng.uniform(0.1, 1.5, n),
                    rng.choice([0.0, 0.02, 0.2], n), rng.choice([0.0, 0.005, 0.05], n)], 1)
    for material in (None, mat):
        world = physics.PhysWorld(b, None, pair_capacity=8 * n, device=cuda_device)
        if material is not None:
            world.set_materials(material)
        world.broadphase()
        world.contacts()
        got, total = world.download_contacts(ob.CONTACT_DTYPE)
        pairs = world.download()["pairs"]
        exp, exp_total = ob.contacts_spheres(pairs, b["pos"], b["radius"], material)
        assert len(got) == len(pairs) and total == exp_total
        assert 0 < exp_total < len(pairs), "AABB overlap without sphere contact exists in the sample"
        assert got.tobytes() == exp.tobytes(), "contact records, every field bit-exact"
        k = int(np.flatnonzero((pairs[:, 0] == 10) & (pairs[:, 1] == 11))[0])
        assert got["nc"][k] == 1 and tuple(got["normal"][k]) == (1.0, 0.0, 0.0)
        if material is not None:
            assert len(np.unique(got["mode"][got["nc"] == 1])) == 2, "with and without dContactBounce"


def test_sphere_box_contacts_match_restatement(cuda_device):
    """(body, static box) narrowphase: dCollideSphereBox + phys_contact_surface, every field bit-exact against
    the restatement, over the statics broadphase's own candidate pairs."""
    from clap_amd import physics
    n = 30_000
    b = synth.sphere_bodies(n, box=24.0, seed=41)
    statics = synth.static_boxes(48, 24.0)
    lo, hi = statics[:, 0::2], statics[:, 1::2]
    b["pos"][5] = (lo[0] + hi[0]) / 2                       # a centre in the middle of a box
    b["pos"][6] = hi[1]                                     # exactly on a corner
    b["pos"][7] = [hi[2, 0], (lo[2, 1] + hi[2, 1]) / 2, (lo[2, 2] + hi[2, 2]) / 2]   # exactly on a face
    rng = np.random.Generator(np.random.PCG64(5))
    mat = np.stack([rng.choice([0.0, 0.3, 0.8], n), rng.uniform(0, 0.2, n), rng.uniform(0.1, 1.5, n),
                    rng.choice([0.0, 0.02, 0.2], n), rng.choice([0.0, 0.005, 0.05], n)], 1)
    smat = np.stack([rng.choice([0.0, 0.5], 48), rng.uniform(0, 0.2, 48), rng.uniform(0.1, 1.5, 48),
                     rng.choice([0.0, 0.1], 48), rng.choice([0.0, 0.03], 48)], 1)
    for material, static_material in ((None, None), (mat, smat)):
        world = physics.PhysWorld(b, statics, pair_capacity=8 * n, device=cuda_device)
        if material is not None:
            world.set_materials(material)
        world.broadphase()
        world.contacts_static(static_material)
        got, total = world.download_static_contacts(ob.CONTACT_DTYPE)
        pairs = world.download()["static_pairs"]
        exp, exp_total = ob.contacts_sphere_box(pairs, b["pos"], b["radius"], statics, material, static_material)
        assert len(got) == len(pairs) and total == exp_total
        assert 0 < exp_total < len(pairs), "sphere-AABB overlap without sphere-box contact exists in the sample"
        assert got.tobytes() == exp.tobytes(), "contact records, every field bit-exact"
        inside = (np.abs(got["normal"]).sum(axis=1) == 1.0) & (got["nc"] == 1)
        assert inside.any() and (~inside & (got["nc"] == 1)).any(), "both branches of dCollideSphereBox were taken"
        if material is not None:
            assert len(np.unique(got["mode"][got["nc"] == 1])) == 2, "with and without dContactBounce"


def test_sphere_contacts_empty_and_truncated(cuda_device):
    from clap_amd import physics
    b = synth.sphere_bodies(3000, box=6.0, seed=6)
    world = physics.PhysWorld(b, None, pair_capacity=100, device=cuda_device)   # fewer slots than pairs found
    world.broadphase()
    world.contacts()
    got, total = world.download_contacts(ob.CONTACT_DTYPE)
    pairs = world.download()["pairs"]
    assert len(got) == len(pairs) == 100
    exp, exp_total = ob.contacts_spheres(pairs, b["pos"], b["radius"])
    assert total == exp_total and got.tobytes() == exp.tobytes()
    far = synth.sphere_bodies(50, box=4000.0, seed=1)       # nothing overlaps: no pairs, no contacts
    world = physics.PhysWorld(far, None, pair_capacity=64, device=cuda_device)
    world.broadphase()
    world.contacts()
    got, total = world.download_contacts(ob.CONTACT_DTYPE)
    assert len(got) == 0 and total == 0


def test_entity_rotation_pushed_to_linked_bodies(cuda_device):
    """default_update -> phys_body_rotate_xform for characters / static colliders (model.c:1680-1687)."""
    from clap_amd import _lib, entities, physics
    n = 300
    scene = synth.pad_levels(synth.entities_chains(120, 3, seed=6))
    scene["flags"] = scene["flags"].copy()
    roots = np.flatnonzero(scene["parent"] < 0)
    child = np.flatnonzero(scene["parent"] >= 0)[:20]
    scene["flags"][roots[::3]] &= ~np.uint32(_lib.E_DIRTY)               # a third of the roots did not move
    b = synth.sphere_bodies(n, box=10.0, seed=6)
    link_entity = np.concatenate([roots[:100], child]).astype(np.uint32)
    link_body = np.arange(len(link_entity), dtype=np.uint32) * 2
    batch = entities.EntityBatch(scene, cuda_device)
    world = physics.PhysWorld(b, None, device=cuda_device)
    dirty = ((scene["flags"] & _lib.E_DIRTY) != 0).astype(np.uint8)
    for all_dirty in (False, True):
        exp = b["quat"].copy()
        ob.bodies_rotate_from_entities(link_body, link_entity, scene["rot"], scene["parent"],
                                       np.ones_like(dirty) if all_dirty else dirty, exp)
        world.quat.copy_(__import__("torch").from_numpy(b["quat"]))
        world.rotate_from_entities(batch, link_body, link_entity, all_dirty=all_dirty)
        got = world.download()["quat"]
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
        changed = (got != b["quat"]).any(axis=1)
        assert changed[link_body[:100]].sum() == (66 if not all_dirty else 100)
        assert not changed[link_body[100:]].any(), "attached entities take the parent branch: no push"
        assert np.allclose(np.linalg.norm(got[changed], axis=1), 1.0, atol=1e-15)


# ---------------------------------------------------------------- round 2: capsule narrowphase + sweep
def _materials(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.stack([rng.choice([0.0, 0.3, 0.8], n), rng.uniform(0, 0.2, n), rng.uniform(0.1, 1.5, n),
                     rng.choice([0.0, 0.02, 0.2], n), rng.choice([0.0, 0.005, 0.05], n)], 1)


def _oracle_body_geoms(b, st, material=None):
    return ob.geoms(b["n"], pos=st["pos"], axis=st["axis"], radius=b["radius"], length=b.get("length"), material=material)


def test_capsule_contacts_match_restatement(cuda_device):
    """near_callback over both candidate lists of a capsule / sphere mix: dCollideCapsuleCapsule (closest-points
    branch and the two-contact parallel branch), dCollideCapsuleSphere in both argument orders (dCollide's reversal),
    dCollideSpheres, dCollideCapsuleBox (incl. the deep-penetration flag), dCollideSphereBox, with the surface
    parameters -- every field of every record bit-exact against the restatement."""
    import torch
    from clap_amd import physics
    n = 20_000
    b = synth.capsule_bodies(n, box=26.0, seed=33)
    caps = np.flatnonzero(b["length"] > 0)
    # parallel neighbours: copies of a capsule's orientation, shifted sideways by less than two radii
    for k in range(0, 40, 2):
        i, j = caps[k], caps[k + 1]
        b["quat"][j] = b["quat"][i]
        b["pos"][j] = b["pos"][i] + [0.05, 0.0, 0.04]
        b["length"][j], b["radius"][j] = b["length"][i], b["radius"][i]
    b["pos"][caps[50]] = b["pos"][caps[51]]                # coincident capsule centres
    statics = synth.static_boxes(48, 26.0)
    lo, hi = statics[:, 0::2], statics[:, 1::2]
    b["pos"][caps[60]] = (lo[3] + hi[3]) / 2               # a capsule whose axis runs through a box: ODE's dBoxBox case
    mat, smat = _materials(n, 4), _materials(48, 5)
    for material, static_material in ((None, None), (mat, smat)):
        world = physics.PhysWorld(b, statics, pair_capacity=16 * n, device=cuda_device)
        if material is not None:
            world.set_materials(material)
            world.static_material = torch.from_numpy(static_material).to(cuda_device)
        world.broadphase()
        world.contacts_geoms()
        got = world.download_contacts2(ob.CONTACT2_DTYPE)
        out = world.download()
        st = ob.bodies_state(b)
        ob.bodies_aabb(b, st)
        A = _oracle_body_geoms(b, st, material)
        S = ob.geoms(48, kind=np.full(48, 2, np.uint8), aabb=statics, material=static_material)
        exp, exp_total = ob.contacts_geoms(out["pairs"], A, A)
        recs, total = got["body"]
        assert len(recs) == len(out["pairs"]) and total == exp_total
        assert recs.tobytes() == exp.tobytes(), "body x body contact records, every field bit-exact"
        assert (exp["nc"] == 2).sum() >= 10, "parallel capsules give two contacts"
        assert 0 < exp_total < len(exp)
        kinds = (b["length"][out["pairs"]] > 0).astype(int)
        touching = exp["nc"] > 0
        for combo in ((0, 0), (0, 1), (1, 0), (1, 1)):
            assert (touching & (kinds[:, 0] == combo[0]) & (kinds[:, 1] == combo[1])).any(), combo
        exp_s, exp_s_total = ob.contacts_geoms(out["static_pairs"], A, S)
        recs_s, total_s = got["static"]
        assert total_s == exp_s_total and recs_s.tobytes() == exp_s.tobytes(), "body x static contact records"
        assert (exp_s["nc"] == 0x80000000).any(), "the deep-penetration capsule is flagged"
        assert ((exp_s["nc"] == 1) & (b["length"][out["static_pairs"][:, 0]] > 0)).any()
        # bodies of a touching pair hold a joint now (the flagged deep pairs are left to the host: no joint here)
        hit = np.zeros(n, bool)
        hit[out["pairs"][np.isin(exp["nc"], (1, 2))].ravel()] = True
        hit[out["static_pairs"][np.isin(exp_s["nc"], (1, 2))][:, 0]] = True
        assert np.array_equal((out["bflags"] & 16) != 0, hit), "CLAPGPU_BODY_HAS_JOINT on exactly the touching bodies"


def test_contacts_of_both_lists_in_one_launch(cuda_device):
    """clapgpu_contacts_geoms_both: the records, totals and joint flags of the two clapgpu_contacts_geoms calls it replaces
    (near_callback over dSpaceCollide's and dSpaceCollide2's pairs, physics.c:751-753), byte for byte -- three frames in a
    row, because its ticket-and-counts word has to be back at zero for the next launch, and once with a pair capacity
    smaller than the list."""
    import torch
    from clap_amd import physics
    n = 30_000
    b = synth.capsule_bodies(n, box=30.0, seed=35)
    statics = synth.static_boxes(48, 30.0)
    mat, smat = _materials(n, 6), _materials(48, 7)
    for cap in (16 * n, 9_000):
        two = physics.PhysWorld(b, statics, pair_capacity=cap, device=cuda_device)
        one = physics.PhysWorld(b, statics, pair_capacity=cap, device=cuda_device)
        for w in (two, one):
            w.set_materials(mat)
            w.static_material = torch.from_numpy(smat).to(cuda_device)
        for frame in range(3):
            for w in (two, one):
                w.world_step(1 / 120)
                w.broadphase()
            two.contacts_geoms()
            one.contacts_geoms_both()
            a, c = two.download_contacts2(ob.CONTACT2_DTYPE), one.download_contacts2(ob.CONTACT2_DTYPE)
            oa, oc = two.download(), one.download()
            assert np.array_equal(oa["pairs"], oc["pairs"]) and np.array_equal(oa["static_pairs"], oc["static_pairs"])
            for which in ("body", "static"):
                assert a[which][1] == c[which][1], f"cap {cap} frame {frame}: {which} contact total {a[which][1]} vs {c[which][1]}"
                assert a[which][0].tobytes() == c[which][0].tobytes(), f"cap {cap} frame {frame}: {which} records"
            assert a["body"][1] > 0 and a["static"][1] > 0
            assert np.array_equal(oa["bflags"], oc["bflags"]), "CLAPGPU_BODY_HAS_JOINT"


@pytest.mark.parametrize("grid", ["3", "100000"])
def test_contact_kernels_with_few_and_many_workgroups(grid, cuda_device, monkeypatch):
    """The contact kernels launch as many workgroups as are resident and a wavefront walks its chunks of 64 pairs with the
    next chunk's pairs in flight; at test sizes a wavefront has one chunk.  CLAPGPU_CONTACTS_GRID (read at every call) = 3
    workgroups makes each of twelve wavefronts walk dozens of chunks of both lists; 100 000 gives every chunk its own
    wavefront: the same records, totals and joint flags as the restatement either way, single-list and one-launch form."""
    from clap_amd import physics
    monkeypatch.setenv("CLAPGPU_CONTACTS_GRID", grid)
    n = 20_000
    b = synth.capsule_bodies(n, box=26.0, seed=41)
    statics = synth.static_boxes(48, 26.0)
    two = physics.PhysWorld(b, statics, pair_capacity=16 * n, device=cuda_device)
    one = physics.PhysWorld(b, statics, pair_capacity=16 * n, device=cuda_device)
    for w in (two, one):
        w.broadphase()
    two.contacts_geoms()
    one.contacts_geoms_both()
    a, c = two.download_contacts2(ob.CONTACT2_DTYPE), one.download_contacts2(ob.CONTACT2_DTYPE)
    out, out1 = two.download(), one.download()
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    A = _oracle_body_geoms(b, st, None)
    S = ob.geoms(48, kind=np.full(48, 2, np.uint8), aabb=statics, material=None)
    exp, exp_total = ob.contacts_geoms(out["pairs"], A, A)
    exp_s, exp_s_total = ob.contacts_geoms(out["static_pairs"], A, S)
    assert len(out["pairs"]) > 64 * 12 * 4, "several chunks per wavefront at grid 3"
    for got in (a, c):
        assert got["body"][1] == exp_total and got["body"][0].tobytes() == exp.tobytes()
        assert got["static"][1] == exp_s_total and got["static"][0].tobytes() == exp_s.tobytes()
    assert np.array_equal(out["bflags"], out1["bflags"])


def test_static_sphere_and_capsule_colliders(cuda_device):
    """Static colliders the reference creates as capsule / sphere geoms (phys_body_new with a geom only, physics.c:
    980-991): narrowphase against their real shape, candidates from their AABBs."""
    from clap_amd import physics
    n, ns = 8000, 300
    b = synth.capsule_bodies(n, box=16.0, seed=3)
    sb = synth.capsule_bodies(ns, box=16.0, seed=77)
    sst = ob.bodies_state(sb)
    ob.bodies_aabb(sb, sst)
    statics = sst["aabb"].copy()
    kind = (sb["length"] > 0).astype(np.uint8)
    world = physics.PhysWorld(b, statics, pair_capacity=16 * n, device=cuda_device)
    world.set_static_geoms(kind, pos=sb["pos"], axis=sst["axis"], radius=sb["radius"], length=sb["length"])
    world.broadphase()
    world.contacts_geoms()
    got = world.download_contacts2(ob.CONTACT2_DTYPE)
    out = world.download()
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    A = _oracle_body_geoms(b, st)
    S = ob.geoms(ns, pos=sb["pos"], axis=sst["axis"], radius=sb["radius"], length=sb["length"], kind=kind, aabb=statics)
    exp_s, tot = ob.contacts_geoms(out["static_pairs"], A, S)
    assert got["static"][1] == tot and got["static"][0].tobytes() == exp_s.tobytes()
    assert 0 < tot < len(exp_s)


def test_capsule_sweeps_match_restatement(cuda_device):
    """phys_body_sweep_capsule for 300 bodies at once: marching probe, contact filtering by direction, back-up
    distance, first-smallest-fraction rule, early exit -- frac, normal and hit equal the restatement's, exactly."""
    from clap_amd import physics
    n = 6000
    b = synth.capsule_bodies(n, box=14.0, seed=19)
    statics = synth.static_boxes(40, 14.0)
    world = physics.PhysWorld(b, statics, device=cuda_device)
    st = ob.bodies_state(b)
    ob.bodies_aabb(b, st)
    rng = np.random.Generator(np.random.PCG64(2))
    movers = rng.choice(n, 300, replace=False).astype(np.uint32)
    delta = rng.normal(0, 1.0, (300, 3)).astype(np.float32)
    delta[:5] = 0                                           # no movement: frac 1
    delta[5:10] *= 1e-3                                     # shorter than a step
    delta[10:40, 1] = -np.abs(delta[10:40, 1]) - 2.0        # straight down onto the ground slab
    bb = st["aabb"]
    cand, first = [], [0]
    for k, m in enumerate(movers):                          # candidates: everything the swept AABB touches
        lo = np.minimum(bb[m, 0::2], bb[m, 0::2] + delta[k]) - 1e-3
        hi = np.maximum(bb[m, 1::2], bb[m, 1::2] + delta[k]) + 1e-3
        s_hit = np.flatnonzero(np.all((statics[:, 0::2] <= hi) & (statics[:, 1::2] >= lo), axis=1))
        b_hit = np.flatnonzero(np.all((bb[:, 0::2] <= hi) & (bb[:, 1::2] >= lo), axis=1))
        cand += list(s_hit.astype(np.uint32)) + list((b_hit.astype(np.uint32) | np.uint32(1 << 31)))
        first.append(len(cand))
    cand = np.asarray(cand, np.uint32)
    frac, normal, hit = world.sweep_capsules(movers, delta, np.asarray(first, np.uint32), cand)
    frac, normal, hit = frac.cpu().numpy(), normal.cpu().numpy(), hit.cpu().numpy()
    A = _oracle_body_geoms(b, st)
    S = ob.geoms(40, kind=np.full(40, 2, np.uint8), aabb=statics)
    blocked = 0
    for k, m in enumerate(movers):
        f, nrm, h = ob.sweep_capsule(A, m, delta[k], S, cand[first[k]:first[k + 1]])
        assert np.float32(f).tobytes() == frac[k].tobytes(), f"sweep {k}: frac {frac[k]} vs {f}"
        assert nrm.tobytes() == normal[k].tobytes() and h == hit[k], f"sweep {k}"
        blocked += f < 1.0
    assert 30 < blocked < 300 and (frac[:5] == 1.0).all()
    assert (hit <= -2).any() and (hit >= 0).any(), "statics and bodies were both hit"


def test_geom_records_give_the_same_contacts_as_the_arrays(cuda_device):
    """clapgpu_bodies.geom_records / clapgpu_geoms.records: the narrowphase's one-sector view of every body geom (position,
    axis, radius, length in 64 bytes, rewritten by the step and AABB kernels and by the character feeder's teleport) is a
    CACHE of the arrays -- the contact records of both lists and the body flags come out byte for byte as without it,
    before and after bodies have moved, and equal the oracle's."""
    from clap_amd import physics
    n = 30_000
    b = synth.capsule_bodies(n, box=30.0, seed=19)
    statics = synth.static_boxes(24, 30.0)
    worlds = [physics.PhysWorld(b, statics, pair_capacity=16 * n, device=cuda_device, geom_records=r) for r in (True, False)]
    assert worlds[0].geom_records is not None and worlds[1].geom_records is None
    for step in range(3):
        outs = []
        for w in worlds:
            w.broadphase()
            w.contacts_geoms()
            outs.append((w.download_contacts2(ob.CONTACT2_DTYPE), w.download()))
        (c0, d0), (c1, d1) = outs
        assert c0["body"][1] == c1["body"][1] > 0 and c0["static"][1] == c1["static"][1]
        assert c0["body"][0].tobytes() == c1["body"][0].tobytes(), f"step {step}: body x body records"
        assert c0["static"][0].tobytes() == c1["static"][0].tobytes(), f"step {step}: body x static records"
        assert np.array_equal(d0["bflags"], d1["bflags"])
        rec = worlds[0].geom_records.cpu().numpy()
        assert np.array_equal(rec[:, 0:3], d0["pos"]) and np.array_equal(rec[:, 3:6], d0["axis"])
        assert np.array_equal(rec[:, 6], b["radius"]) and np.array_equal(rec[:, 7], b["length"])
        for w in worlds:
            w.world_step(1.0 / 120.0)
            w.world_step(1.0 / 120.0)


@pytest.mark.parametrize("kind", ["spheres", "capsules"])
def test_step_that_bins_for_the_next_broadphase_gives_the_same_pairs(kind, cuda_device):
    """clapgpu_bodies_step_prebin: the substep's body step also runs the NEXT broadphase's first launch over the boxes it
    writes (k_bp_bin: cell slot, rank in the cell, the cell counters, the epoch), and clapgpu_bp_collide skips that launch.
    Two worlds from the same bodies -- A: collide, step; B: collide, step + pre-bin -- over eight substeps: body state bit
    for bit, both pair lists identical every substep (the emitted list is canonical whatever order the atomics took).
    Bodies go to sleep on the way (auto-disable: a disabled body is binned from its stored box).  Then boxes rewritten
    behind the pre-binned step's back (clapgpu_bodies_aabb -> clapgpu_bp_invalidate): the next collide bins for itself."""
    import torch
    from clap_amd import physics
    n = 20_000
    b = synth.sphere_bodies(n, box=40.0, seed=21) if kind == "spheres" else synth.capsule_bodies(n, box=40.0, seed=21)
    b["lvel"][: n // 3] *= 1e-4                                           # a third nearly at rest: these fall asleep
    b["avel"][: n // 3] *= 1e-4
    stat = synth.static_boxes(32, 40.0)
    A = physics.PhysWorld(b, stat, pair_capacity=400_000, device=cuda_device)
    B = physics.PhysWorld(b, stat, pair_capacity=400_000, device=cuda_device)
    for w in (A, B):
        w.world.adis_steps = 3                                           # short fuse: disabling happens inside the test
    for s in range(8):
        A.broadphase(); B.broadphase()
        da, db = A.download(), B.download()
        assert da["pair_total"] == db["pair_total"] > 0 and np.array_equal(da["pairs"], db["pairs"]), f"substep {s}: body pairs"
        assert da["static_pair_total"] == db["static_pair_total"] and np.array_equal(da["static_pairs"], db["static_pairs"]), f"substep {s}: static pairs"
        assert A.broadphase_status() == B.broadphase_status() == 0
        if s in (2, 5):                                                  # contacts in between set HAS_JOINT: the auto-disable path runs
            A.alloc_contacts(); B.alloc_contacts()
            A.contacts_geoms_both(); B.contacts_geoms_both()
        A.world_step(1.0 / 120.0)
        B.world_step(1.0 / 120.0, prebin=True)
        da, db = A.download(), B.download()
        for k in ("pos", "quat", "lvel", "avel", "aabb"):
            assert np.array_equal(da[k].view(np.uint64), db[k].view(np.uint64)), f"substep {s}: {k}"
        assert np.array_equal(da["bflags"], db["bflags"])
    # boxes rewritten behind the pre-binned step's back
    B.pos += 0.25
    A.pos += 0.25
    A.bodies_aabb(); B.bodies_aabb()                                     # (B.bodies_aabb invalidates the pre-binned state)
    A.broadphase(); B.broadphase()
    da, db = A.download(), B.download()
    assert da["pair_total"] == db["pair_total"] and np.array_equal(da["pairs"], db["pairs"])
    assert np.array_equal(da["static_pairs"], db["static_pairs"])
    # two pre-binning steps in a row without a collide in between: the second starts over
    B.world_step(1.0 / 120.0, prebin=True); B.world_step(1.0 / 120.0, prebin=True)
    A.world_step(1.0 / 120.0); A.world_step(1.0 / 120.0)
    A.broadphase(); B.broadphase()
    da, db = A.download(), B.download()
    assert da["pair_total"] == db["pair_total"] and np.array_equal(da["pairs"], db["pairs"])
