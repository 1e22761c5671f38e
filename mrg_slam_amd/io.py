"""Wire and on-disk formats at the seam of the scan-matching path.  Host-side plumbing only: every
function produces or consumes the packed N x 4 float32 (x, y, z, intensity) layout the C ABI ingests without repacking
(``stride_bytes = 16``; the ``*_device`` entry points take the same layout already in HBM).

* ``sensor_msgs/PointCloud2`` as the reference's replay scripts build it — four FLOAT32 fields x, y, z, intensity at
  offsets 0/4/8/12, point_step 16 (/root/reference/python_scripts/kitti_singlerobot_processor.py:164-185);
* KITTI velodyne ``.bin`` — float32 N x 4 (:118-121 via pykitti);
* binary PCD as ``pcl::io::savePCDFileBinary`` writes a ``PointXYZI`` keyframe cloud and ``pcl::io::loadPCDFile`` reads it
  (/root/reference/src/mrg_slam/keyframe.cpp:109,196);
* TUM trajectory text, ``sec.nanosec tx ty tz qx qy qz qw`` (src/mrg_slam/graph_database.cpp:633-637 — C++ stream
  formatting, 6 significant digits; python_scripts/odom_to_file.py:41-43 — Python ``str(float)``).
"""
from __future__ import annotations

import os

import numpy as np

_F32 = np.dtype("<f4")
# (numpy kind, PCD TYPE letter) by PCD SIZE
_PCD_TYPES = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}


# ---- sensor_msgs/PointCloud2 ---------------------------------------------------------------------------------------
def xyzi_from_pointcloud2(data, width: int, height: int, point_step: int, fields: dict[str, int], is_bigendian: bool = False) -> np.ndarray:
    """N x 4 float32 from the byte payload of a PointCloud2 whose x, y, z (and optionally intensity) fields are FLOAT32 at
    the byte offsets in ``fields``.  The replay scripts' layout (offsets 0, 4, 8, 12, point_step 16) comes back as a
    zero-copy view of ``data``; anything else is gathered once."""
    if is_bigendian:
        raise ValueError("big-endian PointCloud2 payloads are not produced by the reference's tooling")
    n = int(width) * int(height)
    buf = np.frombuffer(data, dtype=np.uint8, count=n * point_step)
    offs = [fields["x"], fields["y"], fields["z"], fields.get("intensity")]
    if point_step == 16 and offs == [0, 4, 8, 12]:
        return buf.view(_F32).reshape(n, 4)
    rows = buf.reshape(n, point_step)
    out = np.zeros((n, 4), dtype=np.float32)
    for c, o in enumerate(offs):
        if o is not None:
            out[:, c] = rows[:, o:o + 4].copy().view(_F32).reshape(n)
    return out


def pointcloud2_from_xyzi(cloud) -> dict:
    """The PointCloud2 members kitti_to_ros_point_cloud fills (:166-183) for a packed cloud; ``data`` shares the array's memory."""
    c = np.ascontiguousarray(cloud, dtype=np.float32)
    if c.ndim != 2 or c.shape[1] != 4:
        raise ValueError("clouds are N x 4 float32 arrays (x, y, z, intensity)")
    return {"height": 1, "width": len(c), "is_dense": False, "is_bigendian": False, "point_step": 16, "row_step": 16 * len(c),
            "fields": {"x": 0, "y": 4, "z": 8, "intensity": 12}, "data": memoryview(c).cast("B")}


# ---- KITTI ---------------------------------------------------------------------------------------------------------
def read_kitti_bin(path: str) -> np.ndarray:
    if os.path.getsize(path) % 16:
        raise ValueError(f"{path}: size is not a multiple of 16 bytes")
    return np.fromfile(path, dtype=_F32).reshape(-1, 4)


def write_kitti_bin(path: str, cloud) -> None:
    np.ascontiguousarray(cloud, dtype=_F32).reshape(-1, 4).tofile(path)


# ---- PCD -----------------------------------------------------------------------------------------------------------
def write_pcd_binary(path: str, cloud) -> None:
    """What pcl::io::savePCDFileBinary emits for a pcl::PointCloud<pcl::PointXYZI> (unorganised, default viewpoint): the
    v0.7 header PCDWriter::generateHeader produces and the four fields of every point packed to 16 bytes."""
    c = np.ascontiguousarray(cloud, dtype=_F32).reshape(-1, 4)
    n = len(c)
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
              f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(c.tobytes())


def read_pcd(path: str) -> np.ndarray:
    """N x 4 float32 (x, y, z, intensity; intensity 0 when the file has none) from an ``ascii`` or ``binary`` PCD with any
    field order and padding fields.  ``binary_compressed`` (LZF) is not produced by the reference and is rejected."""
    with open(path, "rb") as f:
        raw = f.read()
    meta, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        meta[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    names, sizes, types = meta["FIELDS"], [int(v) for v in meta["SIZE"]], meta["TYPE"]
    counts = [int(v) for v in meta.get("COUNT", ["1"] * len(names))]
    n = int(meta["POINTS"][0]) if "POINTS" in meta else int(meta["WIDTH"][0]) * int(meta["HEIGHT"][0])
    kind = meta["DATA"][0].lower()
    out = np.zeros((n, 4), dtype=np.float32)
    want = {"x": 0, "y": 1, "z": 2, "intensity": 3}
    if kind == "binary":
        dt, off = [], 0
        for nm, sz, ty, ct in zip(names, sizes, types, counts):
            dt.append((f"{nm}@{off}", _PCD_TYPES[(ty.upper(), sz)], (ct,)))
            off += sz * ct
        rec = np.frombuffer(raw, dtype=np.dtype(dt), count=n, offset=pos)
        for (fname, _, _), nm in zip(dt, names):
            if nm in want:
                out[:, want[nm]] = rec[fname][:, 0]
    elif kind == "ascii":
        tab = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2) if n else np.zeros((0, sum(counts)))
        col = 0
        for nm, ct in zip(names, counts):
            if nm in want:
                out[:, want[nm]] = tab[:n, col]
            col += ct
    else:
        raise ValueError(f"{path}: DATA {kind} is not supported (the reference writes DATA binary)")
    return out


# ---- TUM trajectories ----------------------------------------------------------------------------------------------
def quat_from_rot(R) -> np.ndarray:
    """(x, y, z, w) of a rotation matrix, Eigen::Quaterniond(Matrix3d) branch order (trace first, else largest diagonal)."""
    R = np.asarray(R, dtype=np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        s = np.sqrt(t + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0], q[1], q[2] = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return q


def rot_from_quat(q) -> np.ndarray:
    x, y, z, w = (float(v) for v in q)
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def write_tum(path: str, stamps, poses, style: str = "cpp") -> None:
    """``stamps``: (sec, nanosec) pairs; ``poses``: 4 x 4 matrices.  style "cpp" formats the numbers like the C++ stream of
    graph_database.cpp:636-637 (``%g``, 6 significant digits), "python" like odom_to_file.py:43 (``str(float)``)."""
    fmt = (lambda v: f"{v:g}") if style == "cpp" else (lambda v: str(float(v)))
    with open(path, "w") as f:
        for (sec, nsec), T in zip(stamps, poses):
            T = np.asarray(T, dtype=np.float64)
            q = quat_from_rot(T[:3, :3])
            vals = [T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]]
            f.write(f"{int(sec)}.{int(nsec):09d} " + " ".join(fmt(v) for v in vals) + "\n")


def read_tum(path: str):
    """Returns (stamps, poses): stamps as (sec, nanosec) int pairs, poses as 4 x 4 float64 matrices."""
    stamps, poses = [], []
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            ts, *vals = line.split()
            sec, _, frac = ts.partition(".")
            stamps.append((int(sec), int((frac + "000000000")[:9])))
            v = [float(x) for x in vals]
            T = np.eye(4)
            T[:3, :3] = rot_from_quat(v[3:7])
            T[:3, 3] = v[:3]
            poses.append(T)
    return stamps, poses


# ---- device ingest (the C ABI's mrgfe_ingest_pointcloud2) ---------------------------------------------------------
def ingest_pointcloud2(data, width: int, height: int, point_step: int, fields: dict[str, int], row_step: int = 0, ctx=None, dev_ptr: int = 0) -> np.ndarray | None:
    """pcl::fromROSMsg on the GPU: the PointCloud2 payload goes to the device as it is and x, y, z, intensity are gathered
    there (``mrgfe_ingest_pointcloud2``).  Returns the packed N x 4 float32 cloud, or — with ``dev_ptr`` (room for N packed
    points in device memory) — leaves it there and returns None.  Same result as :func:`xyzi_from_pointcloud2`."""
    import ctypes as C

    from ._lib import check, default_context, lib

    ctx = ctx or default_context()
    n = int(width) * int(height)
    buf = np.frombuffer(data, dtype=np.uint8)
    # the C entry point has no length argument (it copies (height - 1) * row_step + width * point_step bytes): a short or malformed payload
    # must fail here, like xyzi_from_pointcloud2 does, not as an out-of-bounds host read
    need = (int(height) - 1) * (int(row_step) or int(width) * int(point_step)) + int(width) * int(point_step) if n else 0
    if len(buf) < need:
        raise ValueError(f"PointCloud2 payload has {len(buf)} bytes, {need} are needed for {width} x {height} points of {point_step} bytes (row_step {row_step})")
    out = None if dev_ptr else np.empty((n, 4), dtype=np.float32)
    oi = fields.get("intensity")
    check(lib().mrgfe_ingest_pointcloud2(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), int(width), int(height), int(point_step), int(row_step), int(fields["x"]), int(fields["y"]),
                                         int(fields["z"]), -1 if oi is None else int(oi),
                                         out.ctypes.data_as(C.POINTER(C.c_float)) if out is not None else None, C.c_void_p(dev_ptr) if dev_ptr else None))
    return out


def pcl_xyzi_records(cloud) -> np.ndarray:
    """The reference's in-memory layout of a cloud: N 32-byte pcl::PointXYZI records (x, y, z, 1.0f; intensity, 3 padding words),
    as uint8 [N, 32] — what ``stride_bytes = MRGFE_LAYOUT_PCL_XYZI`` ingests."""
    c = np.ascontiguousarray(cloud, dtype=_F32).reshape(-1, 4)
    rec = np.zeros((len(c), 8), dtype=_F32)
    rec[:, :3] = c[:, :3]
    rec[:, 3] = 1.0
    rec[:, 4] = c[:, 3]
    return rec.view(np.uint8).reshape(len(c), 32)
