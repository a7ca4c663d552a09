"""CPU: the oracle's pose / palette restatement against golden vectors produced by the
reference's channels_transform + one_joint_transform (model.c), and live when oracle/_ref is
present.  On the same CPU and libm the restatement is bit-exact (so the 1e-5 bar the GPU is
held to is slack, not need)."""
import glob
import os

import numpy as np
import pytest

from clap_amd import synth
from oracle import binding as ob
from oracle import refrun
from helpers import assert_bits_equal

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pose_*.npz")))


def load_pose(path):
    z = np.load(path)
    sk = {k[3:]: z[k] for k in z.files if k.startswith("sk_")}
    sk["nr_joints"] = sk["parent"].shape[0]
    an = {k[3:]: z[k] for k in z.files if k.startswith("an_")}
    an["n_channels"] = an["ch_target"].shape[0]
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    an["time_end"] = float(ref["time_end"][0])
    return sk, an, dict(char_mx=z["in_char_mx"], trs0=z["in_trs0"]), z["in_char_times"], ref


def test_golden_present():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_golden(path):
    sk, an, ch, times, ref = load_pose(path)
    sk["bind"] = ob.skeleton_bind(sk)
    assert_bits_equal(sk["bind"], ref["bind"], "bind = invert(invmx)")
    n, J = ch["char_mx"].shape[0], sk["nr_joints"]
    trs = np.tile(ch["trs0"], (n, 1, 1))
    cur = np.zeros((n, J, 3), np.int32)
    reach = sk["order"]
    unreach = np.setdiff1d(np.arange(J), reach)
    for f in range(times.shape[0]):
        jt, gl, jp = ob.pose(sk, an, times[f], ch["char_mx"], trs, cur)
        assert_bits_equal(trs, ref["trs"][f], f"frame {f} joint T/R/S")
        assert_bits_equal(jt[:, reach], ref["joint_transforms"][f][:, reach], f"frame {f} joint_transforms")
        assert_bits_equal(gl[:, reach], ref["globalmx"][f][:, reach], f"frame {f} joint global")
        assert_bits_equal(jp[:, reach], ref["joint_pos"][f][:, reach], f"frame {f} joint pos")
        assert not ref["joint_transforms"][f][:, unreach].any(), "reference never writes joints outside joint 0's tree"


def test_fixture_covers_slerp_branches():
    """Rotation keys must exercise dot < 0 (negated), dot > 0.9995 (nlerp) and the acos path."""
    sk, an, ch, times, ref = load_pose(GOLDEN[0])
    rot = [c for c in range(an["n_channels"]) if an["ch_path"][c] == 1]
    dots = []
    for c in rot:
        d = an["data"][an["ch_data_off"][c]:an["ch_data_off"][c] + 4 * an["ch_nr"][c]].reshape(-1, 4)
        dots.append(np.sum(d[1:] * d[:-1], axis=1))
    dots = np.concatenate(dots)
    assert (dots < 0).any() and (np.abs(dots) > 0.9995).any() and (np.abs(dots) < 0.9).any()


@pytest.mark.skipif(not refrun.available(), reason="reference build (oracle/_ref) not present")
def test_oracle_matches_live_reference_c3_shape():
    """BASELINE config 3's skeleton shape (64 joints, depth <= 8, 192 channels x 30 keys)."""
    sk = synth.skeleton(64, 8, 3)
    an = synth.animation(64, 30, 2.0, 3)
    ch = synth.characters(30, 64, seed=3)
    ref = refrun.pose(sk, an, ch, ch["phase"][None, :])
    sk["bind"] = ob.skeleton_bind(sk)
    trs = np.tile(ch["trs0"], (30, 1, 1))
    jt, gl, jp = ob.pose(sk, an, ch["phase"], ch["char_mx"], trs)
    assert_bits_equal(trs, ref["trs"][0], "trs")
    assert_bits_equal(jt, ref["joint_transforms"][0], "joint_transforms")
    assert_bits_equal(jp, ref["joint_pos"][0], "joint_pos")


def test_skin_oracle_identities():
    """Skinning has no reference CPU code (parity unpinned); pin what the shader's formula implies:
    identity palette + weights summing to 1 reproduces the vertex; a single full-weight joint is
    one plain mat4 * vec4."""
    mesh = synth.skinned_mesh(300, 8, seed=2)
    ident = np.tile(np.eye(4, dtype=np.float32).reshape(16), (1, 8, 1))
    p, n = ob.skin(mesh, [0], [300], ident)
    np.testing.assert_allclose(p, mesh["position"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(n, mesh["normal"], rtol=0, atol=3e-7)
    pal = synth._rigid_mat4(np.random.Generator(np.random.PCG64(1)), 8)[None]
    m2 = dict(mesh)
    m2["joints"] = np.zeros_like(mesh["joints"]); m2["joints"][:, 0] = 5
    m2["weights"] = np.zeros_like(mesh["weights"]); m2["weights"][:, 0] = 1
    p, n = ob.skin(m2, [0], [300], pal)
    M = pal[0, 5].reshape(4, 4).T.astype(np.float64)
    exp = (M @ np.concatenate([mesh["position"], np.ones((300, 1))], 1).T).T[:, :3]
    np.testing.assert_allclose(p, exp, rtol=1e-6, atol=1e-6)


def test_animation_clock_matches_reference_fixture(golden_dir):
    """animated_update's time base and restart (model.c:1563-1592): ani_time after every frame of the
    reference's own animated_update, and its poses reproduced from the oracle's float frame times."""
    z = np.load(os.path.join(golden_dir, "animclock_frames.npz"))
    sk = {k[3:]: z[k] for k in z.files if k.startswith("sk_")}
    sk["nr_joints"] = sk["parent"].shape[0]
    an = {k[3:]: z[k] for k in z.files if k.startswith("an_")}
    an["n_channels"] = an["ch_target"].shape[0]
    an["time_end"] = float(z["ref_time_end"][0])
    sk["bind"] = ob.skeleton_bind(sk)
    n, J = len(z["clock_start"]), sk["nr_joints"]
    ani = z["clock_start"].astype(np.float64).copy()
    te = np.asarray([an["time_end"]], np.float32)
    trs = np.tile(z["in_trs0"], (n, 1, 1))
    cur = np.zeros((n, J, 3), np.int32)
    reach = sk["order"]
    ended_total = 0
    for f, now in enumerate(z["clock_now"]):
        ft, ended = ob.animation_time(np.zeros(n, np.uint32), te, ani, z["clock_speed"], z["clock_repeat"], now)
        assert np.array_equal(ani.view(np.uint64), z["ref_ani_time"][f].view(np.uint64)), f"frame {f} ani_time"
        cur[ended != 0] = 0                                  # animation_start resets the search cursors
        jt, _gl, _jp = ob.pose(sk, an, ft, z["in_char_mx"], trs, cur)
        assert_bits_equal(jt[:, reach], z["ref_joint_transforms"][f][:, reach], f"frame {f} joint_transforms")
        ended_total += int(ended.sum())
    assert ended_total > n // 2
