"""include/mrgfe_pcl_adapter.hpp against the stand-in PCL interface of tests/adapter_stub (PCL itself is not installed here).

CPU: the adapter is valid C++ against that interface and links with libmrgfe.so — syntax evidence only, it says nothing about
PCL's real headers.  GPU: the reference's call sequences made through a pcl::Registration BASE pointer
(loop_detector.cpp:104,127-144; scan_matching_odometry_component.cpp:403-417) reach the GPU — PCL's non-virtual
getFitnessScore gets its distances from one batched GPU pass and no CPU kd-tree is built."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "adapter_stub")
INC = ["-I" + STUB, "-I" + os.path.join(ROOT, "include")]


def _gxx():
    g = shutil.which("g++")
    if not g:
        pytest.skip("g++ not available")
    return g


def test_adapter_is_valid_cxx_against_the_stub_interface():
    r = subprocess.run([_gxx(), "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror"] + INC + [os.path.join(STUB, "adapter_main.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_is_inert_without_pcl(tmp_path):
    """without PCL's headers on the include path the adapter headers compile to nothing (they ship as source)"""
    src = tmp_path / "t.cpp"
    src.write_text('#include "mrgfe_pcl_adapter.hpp"\n#include "mrgfe_pcl_filters.hpp"\n#ifdef MRGFE_H\n#error "the adapters must not pull in anything without PCL"\n#endif\nint main() { return 0; }\n')
    r = subprocess.run([_gxx(), "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _build(tmp_path):
    from mrg_slam_amd import _lib

    _lib.build()
    exe = str(tmp_path / "adapter_main")
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([_gxx(), "-std=c++17", "-O1"] + INC + [os.path.join(STUB, "adapter_main.cpp"), "-o", exe, "-L" + libdir, "-lmrgfe", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_adapter_compiles_for_the_pcl_113_signature(tmp_path):
    """pcl::search::KdTree::setInputCloud returns bool from PCL 1.13 on (ROS 2 Jazzy): the override follows PCL_VERSION_COMPARE.  The stand-in
    restates 1.12; with its version macro raised and its own base signature changed accordingly the adapter must still compile."""
    import re

    stub = tmp_path / "stub"
    shutil.copytree(STUB, stub)
    pt = stub / "pcl" / "point_types.h"
    pt.write_text(pt.read_text().replace("PCL_VERSION_CALC(1, 12, 1)", "PCL_VERSION_CALC(1, 13, 0)"))
    kd = stub / "pcl" / "search" / "kdtree.h"
    txt = kd.read_text()
    txt = txt.replace("virtual void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) = 0;",
                      "virtual bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) = 0;")
    txt = re.sub(r"void setInputCloud\(const PointCloudConstPtr& cloud, const IndicesConstPtr& = IndicesConstPtr\(\)\) override\s*\{\s*input_ = cloud;\s*\+\+builds\(\);\s*\}",
                 "bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& = IndicesConstPtr()) override { input_ = cloud; ++builds(); return true; }", txt)
    assert "bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& = IndicesConstPtr()) override" in txt
    kd.write_text(txt)
    r = subprocess.run([_gxx(), "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + str(stub), "-I" + os.path.join(ROOT, "include"), os.path.join(STUB, "adapter_main.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_links_with_the_library(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_reference_call_sequences_through_the_base_pointer(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "6000"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adapter check passed" in r.stdout and "FAIL" not in r.stdout
