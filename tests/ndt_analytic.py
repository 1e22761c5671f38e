"""First-principles NDT score, gradient and Hessian in numpy float64 — an INDEPENDENT check of the derivative arithmetic, not a
restatement of pclomp's code: the Gaussian model of Magnusson 2009 (eqs. 6.9, 6.12, 6.13)

    s(p)   = sum_k  -d1 exp(-d2/2 q^T C q),                    q = R(p) x + t(p) - mu_k,   C = Sigma_k^-1
    g_i    = sum_k   d1 d2 e (q^T C dq_i),                      e = exp(-d2/2 q^T C q)
    H_ij   = sum_k   d1 d2 e ( -d2 (q^T C dq_i)(q^T C dq_j) + q^T C ddq_ij + dq_j^T C dq_i )

with the pose derivatives dq_i, ddq_ij taken from products of the elementary rotation matrices R = Rx(a) Ry(b) Rz(c) and their
first / second derivatives — no expanded trigonometric tables (pclomp's j_ang / h_ang, which the oracle and the kernels share),
no float arithmetic, no 4x6 / 24x6 layouts.  Only the voxel neighbourhood (DIRECT7 / DIRECT1 / all 27) and the leaf statistics
(mean, inverse covariance, point count >= 6) are taken as given.  tests/test_oracle_ndt.py holds the oracle against it,
tests/test_gpu_ndt.py the HIP kernels; the tolerance is the float32 level of the reference's per-pair arithmetic."""
import numpy as np


def gauss_constants(resolution, outlier_ratio=0.55):
    c1 = 10 * (1 - outlier_ratio)
    c2 = outlier_ratio / resolution ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


def _rot(axis, a, order):
    """order-th derivative with respect to the angle of the rotation about `axis` by a"""
    c, s = np.cos(a), np.sin(a)
    cs = [(c, s), (-s, c), (-c, -s)][order]  # (cos, sin) and their derivatives
    cc, ss = cs
    d = 1.0 if order == 0 else 0.0
    if axis == 0:
        return np.array([[d, 0, 0], [0, cc, -ss], [0, ss, cc]])
    if axis == 1:
        return np.array([[cc, 0, ss], [0, d, 0], [-ss, 0, cc]])
    return np.array([[cc, -ss, 0], [ss, cc, 0], [0, 0, d]])


def rotation_derivatives(angles):
    """R, dR[i], ddR[i][j] for R = Rx(a) Ry(b) Rz(c)"""
    def prod(orders):
        return _rot(0, angles[0], orders[0]) @ _rot(1, angles[1], orders[1]) @ _rot(2, angles[2], orders[2])

    R = prod([0, 0, 0])
    dR = [prod([int(i == k) for k in range(3)]) for i in range(3)]
    ddR = [[prod([int(i == k) + int(j == k) for k in range(3)]) for j in range(3)] for i in range(3)]
    return R, dR, ddR


def neighbours(xt, search, leaf, min_b, max_b, div_b, key_to_leaf):
    ijk = np.floor(xt / leaf).astype(np.int64)
    if search == "DIRECT1":
        offs = [(0, 0, 0)]
    elif search == "DIRECT7":
        offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    else:
        offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
    mul = np.array([1, div_b[0], div_b[0] * div_b[1]], dtype=np.int64)
    out = []
    for o in offs:
        c = ijk + np.array(o)
        if (c < min_b).any() or (c > max_b).any():
            continue
        l = key_to_leaf.get(int(((c - min_b) * mul).sum()))
        if l is not None:
            out.append(l)
    return out


def evaluate(source_xyz, p, search, resolution, grid, leaves, outlier_ratio=0.55, transformed=None, upstream_d1_sign=False, nb_lists=None):
    """score, gradient[6], Hessian[6, 6] at pose vector p = (tx, ty, tz, rx, ry, rz).  `transformed` (N x 3 float32): where the
    reference puts the points — it transforms the cloud with its float matrix in float arithmetic and stores float points, and a
    thin (planar) voxel has inverse-covariance eigenvalues above 1000 / m^2, so the last bit of those floats (2e-6 m) already moves
    the Hessian by 1e-4 relative; the model is evaluated AT those points.  The derivatives of the point with respect to the pose
    use p and the untransformed point.
    upstream_d1_sign: PCL / ndt_omp's table of second derivatives has +sin(ry) where d^2 R / d ry^2 has -sin(ry) (row x, column z:
    oracle/quirks.h kNdtHAngD1ZSign); True reproduces that one sign, so that everything ELSE is held against first principles.
    nb_lists: per point the leaves to use instead of the voxel neighbourhood `search` (pcl::NormalDistributionsTransform's radius search: the caller
    finds them by brute force over the centroids, tests/test_oracle_pclndt.py)."""
    min_b, max_b, div_b = (np.asarray(a, dtype=np.int64) for a in grid)
    keys, npts, mean, icov = leaves
    key_to_leaf = {int(k): i for i, k in enumerate(keys) if npts[i] >= 6}
    d1, d2 = gauss_constants(resolution, outlier_ratio)
    p = np.asarray(p, dtype=np.float64)
    R, dR, ddR = rotation_derivatives(p[3:])
    if upstream_d1_sign:
        ddR[1][1] = ddR[1][1].copy()
        ddR[1][1][0, 2] = +np.sin(p[4])  # the true entry is -sin(ry)
    s, g, H = 0.0, np.zeros(6), np.zeros((6, 6))
    for n, x in enumerate(np.asarray(source_xyz, dtype=np.float64)):
        xt = R @ x + p[:3] if transformed is None else np.asarray(transformed[n], dtype=np.float64)
        nb = nb_lists[n] if nb_lists is not None else neighbours(xt.astype(np.float32).astype(np.float64), search, resolution, min_b, max_b, div_b, key_to_leaf)
        if not nb:
            continue
        dq = np.zeros((6, 3))
        dq[:3] = np.eye(3)
        for i in range(3):
            dq[3 + i] = dR[i] @ x
        ddq = np.zeros((6, 6, 3))
        for i in range(3):
            for j in range(3):
                ddq[3 + i, 3 + j] = ddR[i][j] @ x
        for l in nb:
            q = xt - mean[l]
            C = icov[l]
            e = np.exp(-d2 / 2 * q @ C @ q)
            if not (0 <= d2 * e <= 1):
                continue
            s += -d1 * e
            qCd = dq @ (C @ q)
            g += d1 * d2 * e * qCd
            H += d1 * d2 * e * (-d2 * np.outer(qCd, qCd) + ddq @ (C @ q) + dq @ C @ dq.T)
    return s, g, H
