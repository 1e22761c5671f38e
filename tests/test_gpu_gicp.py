"""GPU: GICP_HIP against the CPU oracle (restated fast_gicp::FastGICP) through the C ABI."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _pair(n=3000, seed=0):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt = small_cloud(n, seed)
    rel = synth.make_pose([0.2, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
    src = orc.transform_points(np.linalg.inv(rel), small_cloud(n, seed + 50))
    return tgt, src, rel


def _rot_angle(Ra, Rb):
    from mrg_slam_amd import synth

    return synth.rotation_angle(Ra, Rb)


@pytest.mark.parametrize("eps", [0.1, 0.01])
def test_gicp_align_matches_oracle(eps):
    from mrg_slam_amd import GicpHip
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    g = GicpHip(transformation_epsilon=eps)
    o = orc.FastGicp(transformation_epsilon=eps, num_threads=4)
    g.setInputTarget(tgt)
    o.setInputTarget(tgt)
    g.setInputSource(src)
    o.setInputSource(src)
    aligned = g.align(np.eye(4), want_aligned=True)
    o.align(np.eye(4))
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged()
    assert g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4
    assert _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    np.testing.assert_allclose(g.getHessian(), o.getFinalHessian(), rtol=0, atol=1e-6 * np.abs(o.getFinalHessian()).max())
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-3)
    assert np.linalg.norm(Tg[:3, 3] - rel[:3, 3]) < 0.1
