"""Cloud layouts behind the C ABI: the reference's 32-byte pcl::PointXYZI records and arbitrary
sensor_msgs/PointCloud2 payloads are gathered on the device; a bare stride that cannot say where the intensity lives is refused
(stride_bytes = 32 used to read the 1.0f padding word as the intensity)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_fp = C.POINTER(C.c_float)


def _scan(n=20000, seed=3):
    rng = np.random.default_rng(seed)
    c = np.empty((n, 4), dtype=np.float32)
    c[:, :3] = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
    c[:, 2] *= 0.1
    c[:, 3] = rng.uniform(0.05, 0.95, n).astype(np.float32)  # non-trivial intensities, never 1.0
    return c


def test_pcl_xyzi_layout_through_prefilter_keeps_intensity():
    from mrg_slam_amd import _lib, io
    from mrg_slam_amd._lib import LAYOUT_PCL_XYZI, check, default_context, lib
    from oracle import oracle as orc

    c = _scan()
    rec = io.pcl_xyzi_records(c)
    ctx = default_context()
    q = _lib.PrefilterParams()
    lib().mrgfe_prefilter_default_params(C.byref(q))
    out, m = np.empty((len(c), 4), dtype=np.float32), C.c_size_t(0)
    check(lib().mrgfe_prefilter(ctx._h, C.byref(q), rec.ctypes.data_as(_fp), len(c), LAYOUT_PCL_XYZI, out.ctypes.data_as(_fp), C.byref(m)))
    got = out[: m.value]
    exp = orc.distance_filter(c, 0.1, 35.0)
    exp, _ = orc.voxelgrid(exp, 0.1, 1)
    exp, _ = orc.radius_outlier(exp, 0.5, 2)
    assert got.shape == exp.shape and (got == exp).all()
    assert not np.all(got[:, 3] == 1.0)


def test_pcl_xyzi_layout_registration_entry_points():
    """every stride_bytes entry point decodes the same descriptor: target / source of a registration give the packed call's result"""
    from mrg_slam_amd import NdtHip, io, synth
    from mrg_slam_amd._lib import LAYOUT_PCL_XYZI, check, lib

    scene = synth.street_scene()
    tgt, src, rel = synth.scan_pair(0, "VLP16", scene)
    a, b = NdtHip(transformation_epsilon=0.01), NdtHip(transformation_epsilon=0.01)
    a.setInputTarget(tgt)
    a.setInputSource(src)
    a.align(rel)
    rt, rs = io.pcl_xyzi_records(tgt), io.pcl_xyzi_records(src)
    check(lib().mrgfe_reg_set_target(b._h, rt.ctypes.data_as(_fp), len(tgt), LAYOUT_PCL_XYZI))
    check(lib().mrgfe_reg_set_source(b._h, rs.ctypes.data_as(_fp), len(src), LAYOUT_PCL_XYZI))
    b._n_src = len(src)
    b.align(rel)
    assert (a.getFinalTransformation() == b.getFinalTransformation()).all()
    assert a.getFitnessScore() == b.getFitnessScore()
    # the aligned cloud carries the source intensities
    out = b.align(rel, want_aligned=True)
    assert (out[:, 3] == src[:, 3]).all()


def test_bare_wide_stride_is_refused():
    from mrg_slam_amd import _lib
    from mrg_slam_amd._lib import default_context, lib

    c = _scan(100)
    rec = np.zeros((100, 8), dtype=np.float32)
    ctx = default_context()
    out, m = np.empty((100, 4), dtype=np.float32), C.c_size_t(0)
    st = lib().mrgfe_distance_filter(ctx._h, rec.ctypes.data_as(_fp), 100, 32, 0.1, 35.0, out.ctypes.data_as(_fp), C.byref(m))
    assert st == _lib.ERR_INVALID and b"MRGFE_LAYOUT" in lib().mrgfe_last_error()
    for bad in (_lib.layout(32, 0, 30), _lib.layout(32, 24, 16), _lib.layout(18, 0, 12), _lib.layout(32, 0, 8), 12):
        assert lib().mrgfe_distance_filter(ctx._h, rec.ctypes.data_as(_fp), 100, bad, 0.1, 35.0, out.ctypes.data_as(_fp), C.byref(m)) == _lib.ERR_INVALID
    # 0 and 16 are the packed layout
    for ok in (0, 16, _lib.layout(16, 0, 12)):
        assert lib().mrgfe_distance_filter(ctx._h, c.ctypes.data_as(_fp), 100, ok, 0.1, 35.0, out.ctypes.data_as(_fp), C.byref(m)) == 0


@pytest.mark.parametrize("layout", [
    dict(point_step=16, x=0, y=4, z=8, intensity=12),           # the replay scripts' layout: plain copy
    dict(point_step=32, x=0, y=4, z=8, intensity=16),           # pcl::PointXYZI as toROSMsg serialises it
    dict(point_step=24, x=4, y=8, z=12, intensity=20),          # leading timestamp word
    dict(point_step=20, x=8, y=0, z=4, intensity=16),           # fields out of order
    dict(point_step=12, x=0, y=4, z=8, intensity=None),         # no intensity field
    dict(point_step=16, x=0, y=4, z=8, intensity=12, height=7, pad=48),  # organised cloud with row padding
])
def test_ingest_pointcloud2_matches_host_gather(layout):
    import torch

    from mrg_slam_amd import io

    rng = np.random.default_rng(5)
    height = layout.get("height", 1)
    width = 3000 // height
    n = width * height
    ps = layout["point_step"]
    row_step = width * ps + layout.get("pad", 0)
    data = rng.integers(0, 256, size=height * row_step, dtype=np.uint8)  # garbage everywhere the fields are not
    c = _scan(n, seed=9)
    rows = data.reshape(height, row_step)
    for r in range(height):
        pts = rows[r, : width * ps].reshape(width, ps)
        for col, name in enumerate(("x", "y", "z", "intensity")):
            o = layout[name]
            if o is not None:
                pts[:, o:o + 4] = c[r * width:(r + 1) * width, col].copy().view(np.uint8).reshape(width, 4)
    fields = {k: layout[k] for k in ("x", "y", "z", "intensity")}
    got = io.ingest_pointcloud2(data, width, height, ps, fields, row_step=row_step)
    exp = c.copy()
    if layout["intensity"] is None:
        exp[:, 3] = 0.0
    assert (got == exp).all()
    if "pad" not in layout:
        host = io.xyzi_from_pointcloud2(data, width, height, ps, fields)
        assert (host == exp).all()
    # device output feeds the *_device entry points
    d = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    assert io.ingest_pointcloud2(data, width, height, ps, fields, row_step=row_step, dev_ptr=d.data_ptr()) is None
    assert (d.cpu().numpy() == exp).all()


def test_ingest_pointcloud2_rejects_bad_layouts():
    from mrg_slam_amd import _lib, io

    data = np.zeros(20 * 10, dtype=np.uint8)
    for fields, ps in (({"x": 0, "y": 4, "z": 8, "intensity": 14}, 20), ({"x": 0, "y": 4, "z": 16, "intensity": 12}, 16), ({"x": 0, "y": 4, "z": 8, "intensity": 12}, 18)):
        with pytest.raises(_lib.MrgfeError):
            io.ingest_pointcloud2(data, 5, 1, ps, fields)
    assert io.ingest_pointcloud2(data, 0, 1, 16, {"x": 0, "y": 4, "z": 8, "intensity": 12}).shape == (0, 4)
