"""CPU: pin the oracle's small linear algebra against numpy (the Eigen routines it stands in for are absent)."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_svd6_solve_matches_numpy():
    rng = np.random.default_rng(1)
    for trial in range(50):
        A = rng.normal(size=(6, 6))
        if trial % 2:
            A = A + A.T  # NDT Hessians are (nearly) symmetric, possibly indefinite
        A *= 10.0 ** rng.uniform(-3, 5)
        b = rng.normal(size=6)
        x, s = orc.svd6_solve(A, b)
        np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-8 * np.linalg.cond(A), atol=0)


def test_svd6_rank_deficient_is_pseudo_inverse():
    rng = np.random.default_rng(2)
    B = rng.normal(size=(6, 4))
    A = B @ B.T  # rank 4
    b = rng.normal(size=6)
    x, s = orc.svd6_solve(A, b)
    assert (s[4:] < 1e-12 * s[0]).all()
    np.testing.assert_allclose(x, np.linalg.pinv(A, rcond=1e-12) @ b, rtol=1e-8, atol=1e-10)


def test_svd6_nonfinite_gives_nan():
    A = np.eye(6)
    A[2, 3] = np.nan
    x, _ = orc.svd6_solve(A, np.ones(6))
    assert np.isnan(x).all()


def test_sym_eig3_matches_numpy():
    rng = np.random.default_rng(3)
    for _ in range(100):
        B = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-3, 2)
        A = B @ B.T
        w, V = orc.sym_eig3(A)
        np.testing.assert_allclose(w, np.linalg.eigvalsh(A), rtol=1e-10, atol=1e-14 * np.abs(A).max())
        np.testing.assert_allclose(V @ np.diag(w) @ V.T, A, rtol=0, atol=1e-12 * np.abs(A).max())
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-12)


def test_pose_matrix_euler_roundtrip():
    rng = np.random.default_rng(4)
    for _ in range(100):
        p = np.concatenate([rng.uniform(-20, 20, 3), rng.uniform(0.01, 1.2, 3)])  # positive x-angle: principal branch
        T = orc.pose_to_matrix(p)
        Rx = lambda a: np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = lambda a: np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rz = lambda a: np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        np.testing.assert_allclose(T[:3, :3], Rx(p[3]) @ Ry(p[4]) @ Rz(p[5]), atol=5e-7)
        np.testing.assert_allclose(T[:3, 3], p[:3].astype(np.float32), atol=0)
        np.testing.assert_allclose(orc.euler_xyz(T), p[3:], atol=5e-6)


def test_euler_negative_x_angle_uses_eigen_branch():
    # Eigen's eulerAngles(0,1,2) keeps the first angle in [0, pi]: a small negative x rotation comes back as an
    # equivalent triple near (pi, pi, pi) - and must rebuild the same rotation.
    p = np.array([0, 0, 0, -0.05, 0.02, 0.1])
    T = orc.pose_to_matrix(p)
    e = orc.euler_xyz(T).astype(np.float64)
    assert e[0] > 3.0
    T2 = orc.pose_to_matrix(np.concatenate([np.zeros(3), e]))
    np.testing.assert_allclose(T2[:3, :3], T[:3, :3], atol=2e-6)
