"""CPU: the product's host-side frustum math (libclapgpu's clapgpu_view_matrix /
clapgpu_perspective / clapgpu_frustum_calc run on the host, no GPU needed) against the
golden vectors produced by the reference's subview_calc_frustum."""
import glob
import os

import pytest

from clap_amd import entities
from helpers import assert_bits_equal, load_golden

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "entities_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_host_frustum_matches_reference(path):
    _scene, cam, ref, _frames = load_golden(path)
    fr, view, proj = entities.view_calc_frustum(cam)
    planes, corners = entities.frustum_arrays(fr)
    assert_bits_equal(view, ref["view_mx"], "view_mx")
    assert_bits_equal(proj, ref["proj_mx"], "proj_mx")
    assert_bits_equal(planes, ref["planes"], "planes")
    assert_bits_equal(corners, ref["corners"], "corners")
