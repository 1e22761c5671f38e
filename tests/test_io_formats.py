"""Wire / on-disk formats at the seam: PointCloud2 payloads as kitti_singlerobot_processor.py:164-185
builds them, KITTI .bin, binary PCD keyframes (keyframe.cpp:109,196), TUM trajectories (graph_database.cpp:633-637,
odom_to_file.py:41-43).  Host-side only."""
import numpy as np
import pytest

from mrg_slam_amd import io as mio
from mrg_slam_amd import synth


def _cloud(n=257, seed=1):
    c = np.random.default_rng(seed).normal(0, 20, (n, 4)).astype(np.float32)
    c[:, 3] = np.random.default_rng(seed + 1).uniform(0, 1, n).astype(np.float32)
    return c


def test_pointcloud2_replay_layout_is_zero_copy():
    c = _cloud()
    msg = mio.pointcloud2_from_xyzi(c)
    assert msg["point_step"] == 16 and msg["row_step"] == 16 * len(c) and msg["fields"] == {"x": 0, "y": 4, "z": 8, "intensity": 12}
    assert bytes(msg["data"]) == c.tobytes()  # velo.tobytes() of the replay script
    back = mio.xyzi_from_pointcloud2(msg["data"], msg["width"], msg["height"], msg["point_step"], msg["fields"])
    assert np.shares_memory(back, c) and np.array_equal(back, c)


def test_pointcloud2_padded_layout_is_gathered():
    """pcl::toROSMsg of a PointXYZI cloud: 32-byte points, intensity at offset 16."""
    c = _cloud(100)
    rec = np.zeros((len(c), 8), dtype=np.float32)
    rec[:, 0:3], rec[:, 3], rec[:, 4] = c[:, :3], 1.0, c[:, 3]
    back = mio.xyzi_from_pointcloud2(rec.tobytes(), len(c), 1, 32, {"x": 0, "y": 4, "z": 8, "intensity": 16})
    assert np.array_equal(back, c)
    noint = mio.xyzi_from_pointcloud2(rec.tobytes(), len(c), 1, 32, {"x": 0, "y": 4, "z": 8})
    assert np.array_equal(noint[:, :3], c[:, :3]) and not noint[:, 3].any()


def test_kitti_bin_round_trip(tmp_path):
    c = _cloud(1000)
    p = tmp_path / "000000.bin"
    mio.write_kitti_bin(str(p), c)
    assert p.stat().st_size == 16 * len(c)
    assert np.array_equal(mio.read_kitti_bin(str(p)), c)
    (tmp_path / "bad.bin").write_bytes(b"123")
    with pytest.raises(ValueError):
        mio.read_kitti_bin(str(tmp_path / "bad.bin"))


def test_pcd_binary_is_what_pcl_writes_and_reads_back(tmp_path):
    c = _cloud(300)
    p = tmp_path / "kf.pcd"
    mio.write_pcd_binary(str(p), c)
    raw = p.read_bytes()
    header, _, body = raw.partition(b"DATA binary\n")
    assert header.decode().splitlines() == ["# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7", "FIELDS x y z intensity", "SIZE 4 4 4 4", "TYPE F F F F",
                                            "COUNT 1 1 1 1", "WIDTH 300", "HEIGHT 1", "VIEWPOINT 0 0 0 1 0 0 0", "POINTS 300"]
    assert body == c.tobytes()
    assert np.array_equal(mio.read_pcd(str(p)), c)


def test_pcd_reader_handles_field_order_padding_and_ascii(tmp_path):
    c = _cloud(50)
    # intensity first, a padding field, doubles for z
    dt = np.dtype([("intensity", "<f4"), ("_", "u1", (4,)), ("x", "<f4"), ("y", "<f4"), ("z", "<f8")])
    rec = np.zeros(len(c), dtype=dt)
    rec["intensity"], rec["x"], rec["y"], rec["z"] = c[:, 3], c[:, 0], c[:, 1], c[:, 2].astype(np.float64)
    head = "VERSION 0.7\nFIELDS intensity _ x y z\nSIZE 4 1 4 4 8\nTYPE F U F F F\nCOUNT 1 4 1 1 1\nWIDTH 50\nHEIGHT 1\nPOINTS 50\nDATA binary\n"
    (tmp_path / "a.pcd").write_bytes(head.encode() + rec.tobytes())
    assert np.array_equal(mio.read_pcd(str(tmp_path / "a.pcd")), c)
    lines = "\n".join(" ".join(repr(float(v)) for v in row) for row in c[:, :3])
    (tmp_path / "b.pcd").write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 50\nHEIGHT 1\nPOINTS 50\nDATA ascii\n" + lines + "\n")
    got = mio.read_pcd(str(tmp_path / "b.pcd"))
    assert np.array_equal(got[:, :3], c[:, :3]) and not got[:, 3].any()
    (tmp_path / "c.pcd").write_text("VERSION 0.7\nFIELDS x\nSIZE 4\nTYPE F\nCOUNT 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n")
    with pytest.raises(ValueError):
        mio.read_pcd(str(tmp_path / "c.pcd"))


def test_tum_round_trip_and_formatting(tmp_path):
    poses = synth.arc_trajectory(5, step=1.0)
    poses[3] = synth.make_pose([1.0, -2.5, 0.25], synth.rot_xyz(3.0, 0.2, -2.9))  # w < 0 branch territory
    stamps = [(1317384506 + k, 7 * 10 ** (k + 2)) for k in range(5)]
    for style, tol in (("cpp", 5e-5), ("python", 1e-12)):
        p = tmp_path / f"traj_{style}.txt"
        mio.write_tum(str(p), stamps, poses, style)
        first = p.read_text().splitlines()[0].split()
        assert first[0] == "1317384506.000000700" and len(first) == 8
        s2, p2 = mio.read_tum(str(p))
        assert s2 == stamps
        for A, B in zip(poses, p2):
            assert np.abs(A - B).max() < tol * max(1.0, np.abs(A[:3, 3]).max())
    # the C++ stream prints 6 significant digits
    assert mio.read_tum(str(tmp_path / "traj_cpp.txt"))[1][3][1, 3] == -2.5
    q = mio.quat_from_rot(poses[3][:3, :3])
    assert abs(np.linalg.norm(q) - 1) < 1e-12 and np.allclose(mio.rot_from_quat(q), poses[3][:3, :3], atol=1e-12)


def test_bench_kitti_root_hook(tmp_path, monkeypatch):
    """bench.py's optional KITTI_ROOT input: velodyne .bin scans plus lidar-frame ground truth
    inv(Tr) * pose * Tr from poses/00.txt and calib.txt (kitti_singlerobot_processor.py:95-98)."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seq = tmp_path / "sequences" / "00"
    (seq / "velodyne").mkdir(parents=True)
    (tmp_path / "poses").mkdir()
    clouds = [_cloud(300 + k, 50 + k) for k in range(3)]
    for k, c in enumerate(clouds):
        mio.write_kitti_bin(str(seq / "velodyne" / f"{k:06d}.bin"), c)
    Tr = synth.make_pose([0.27, -0.08, -0.05], synth.rot_xyz(-1.57, 0.02, -1.55))
    lidar = [synth.make_pose([1.0 * k, 0.1 * k, 0.0], synth.rot_z(0.02 * k)) for k in range(3)]
    cam = [Tr @ L @ np.linalg.inv(Tr) for L in lidar]
    np.savetxt(tmp_path / "poses" / "00.txt", np.array([c[:3, :].reshape(12) for c in cam]))
    (seq / "calib.txt").write_text("P0: " + " ".join(["0"] * 12) + "\nTr: " + " ".join(repr(float(v)) for v in Tr[:3, :].reshape(12)) + "\n")
    monkeypatch.setenv("KITTI_ROOT", str(tmp_path))
    scene, poses, scans = bench.make_workload(2, 4, 0, "distance")
    assert scene is None and len(scans) == 3 and len(poses) == 3
    for k in range(3):
        assert np.array_equal(scans[k], clouds[k])
        assert np.allclose(poses[k], lidar[k], atol=1e-9)
    monkeypatch.delenv("KITTI_ROOT")
    scene, poses, scans = bench.make_workload(1, 2, 0, "distance")
    assert scene is not None and len(scans) == 2 and scans[0].shape[1] == 4


def test_ingest_pointcloud2_refuses_a_short_payload():
    """The C entry point has no length argument: the wrapper must refuse a payload that is shorter than its header says (an exception, not
    an out-of-bounds host read) before it reaches the library — no GPU needed to get that far."""
    from mrg_slam_amd import io

    data = np.zeros(10 * 16 - 1, dtype=np.uint8).tobytes()
    with pytest.raises(ValueError, match="payload"):
        io.ingest_pointcloud2(data, 10, 1, 16, {"x": 0, "y": 4, "z": 8, "intensity": 12}, ctx=object())
    with pytest.raises(ValueError, match="payload"):
        io.ingest_pointcloud2(np.zeros(3 * 64, np.uint8).tobytes(), 2, 4, 16, {"x": 0, "y": 4, "z": 8}, row_step=64, ctx=object())
