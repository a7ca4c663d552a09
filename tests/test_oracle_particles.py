"""CPU: the oracle's particle restatement (drand48 stream, respawn, advect, billboard)
against golden vectors produced by the reference's own particle.c, and live when the
reference build is present."""
import glob
import os

import numpy as np
import pytest

from clap_amd import synth
from oracle import binding as ob
from oracle import refrun
from helpers import assert_bits_equal

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "particles_*.npz")))


def load_particles(path):
    z = np.load(path)
    ps = dict(sys=z["in_sys"], row_sys=z["in_row_sys"], n=int(z["in_n"][0]), n_real=int(z["in_n"][1]))
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    return ps, z["in_view_mx"], int(z["in_rng_state"][0]), ref


def test_golden_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_golden(path):
    ps, view, state, ref = load_particles(path)
    pos, vel, st = ob.particles_spawn(ps, state)
    assert_bits_equal(pos, ref["pos0"], "spawn pos")
    assert_bits_equal(vel, ref["vel0"], "spawn vel")
    assert st == int(ref["rng_state"][0])
    respawns = 0
    for f in range(ref["pos"].shape[0]):
        k, st = ob.particles_update(ps, pos, vel, st)
        respawns += k
        assert_bits_equal(pos, ref["pos"][f], f"frame {f} pos_array")
        assert_bits_equal(vel, ref["vel"][f], f"frame {f} velocity")
        assert st == int(ref["rng_state"][f + 1]), f"frame {f} drand48 state"
        for s in range(ps["sys"].shape[0]):
            assert_bits_equal(ob.particles_billboard(view, ps["sys"]["center"][s]), ref["mx"][f][s], "billboard mx")
    assert respawns > 0, "fixture must exercise the respawn branch"


def test_drand48_known_answers():
    """glibc drand48 from its documented default state: first values of the stream."""
    st = ob.C.c_uint64(synth.DRAND48_DEFAULT_STATE)
    vals = [ob.lib().clapo_drand48(ob.C.byref(st)) for _ in range(3)]
    # X1 = (0x5DEECE66D * 0x1234ABCD330E + 0xB) mod 2^48, value = X1 / 2^48
    x = synth.DRAND48_DEFAULT_STATE
    exp = []
    for _ in range(3):
        x = (0x5DEECE66D * x + 0xB) & ((1 << 48) - 1)
        exp.append(x / float(1 << 48))
    assert vals == exp
    assert abs(vals[0] - 0.0) >= 0 and 0 <= min(vals) and max(vals) < 1
    assert ob.lib().clapo_srand48(42) == ((42 << 16) | 0x330E)


@pytest.mark.skipif(not refrun.available(), reason="reference build (oracle/_ref) not present")
def test_oracle_matches_live_reference_many_systems():
    ps = synth.particle_systems(n_sys=40, count=700, radius=5.0, velocity=0.8, ragged=True, seed=77)
    view = np.eye(4, dtype=np.float32).ravel()
    ref = refrun.particles(ps, view, 0x123456789ABC, 4)
    pos, vel, st = ob.particles_spawn(ps, 0x123456789ABC)
    assert_bits_equal(pos, ref["pos0"], "spawn")
    for f in range(4):
        _k, st = ob.particles_update(ps, pos, vel, st)
        assert_bits_equal(pos, ref["pos"][f], f"frame {f}")
        assert st == int(ref["rng_state"][f + 1])


def test_harness_spawn_matches_oracle_bits():
    """clap_amd.synth.particles_spawn (numpy, drand48 by jump-ahead: what bench.py and the tools prepare device arrays
    with) against the oracle's sequential spawn: positions, velocities and the stream state, bit for bit, for every
    radius distribution and for ragged systems."""
    for kw in (dict(n_sys=7, count=100, ragged=True, seed=9), dict(n_sys=40, count=1024), dict(n_sys=1, count=1),
               dict(n_sys=9, count=77, dist=synth.PART_DIST_CBRT), dict(n_sys=9, count=77, dist=synth.PART_DIST_POW075),
               dict(n_sys=9, count=77, dist=synth.PART_DIST_LIN, min_radius=2.5)):
        ps = synth.particle_systems(**kw)
        for state in (synth.DRAND48_DEFAULT_STATE, 1, (1 << 48) - 1):
            a = synth.particles_spawn(ps, state)
            b = ob.particles_spawn(ps, state)
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), kw
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), kw
            assert a[2] == b[2], kw
    vals, st = synth.drand48_stream(synth.DRAND48_DEFAULT_STATE, 200_001)
    s = synth.DRAND48_DEFAULT_STATE
    for k in range(200_001):
        s = (0x5DEECE66D * s + 0xB) & ((1 << 48) - 1)
        if k in (0, 1, 65535, 65536, 131071, 200_000):
            assert vals[k] == s / 2.0 ** 48, k
    assert st == s
