"""The synthetic measurement inputs (mrg_slam_amd/synth.py) are the same BITS on every host.

Record digests of bench.py's workloads are only comparable across machines when the inputs are (two GPU boxes
ray-cast different last bits with numpy's SIMD sin / cos and BLAS products).  The generator is now built from IEEE + - * / sqrt on
float64 arrays alone; this file pins one digest over scenes, trajectories, scans of both sensor models and perturbed guesses.  The same
check runs on the GPU box (`-m gpu` copy below: another CPU model) and bench.py prints the digests of its own workloads."""
import hashlib

import numpy as np
import pytest

from mrg_slam_amd import synth

PINNED = "5ee96959d57393d0c810c3fb17aa0d36841cc1505c0546b0be624729740b1efd"  # computed in the build container (Xeon, AVX-512)


def digest() -> str:
    h = hashlib.sha256()
    sc = synth.street_scene(seed=7, x_range=(-60.0, 80.0))
    h.update(sc.boxes.tobytes())
    h.update(sc.cylinders.tobytes())
    ls = synth.loop_scene(radius=30.0)
    h.update(ls.boxes.tobytes())
    h.update(ls.cylinders.tobytes())
    poses = synth.weave_trajectory(5) + synth.loop_trajectory(4, 30.0) + synth.arc_trajectory(3)
    for P in poses:
        h.update(P.tobytes())
    for k, P in enumerate(poses[:5]):
        h.update(synth.synth_lidar(sc, P, "VLP16", synth.BASE_SEED + k, azimuth_steps=360).tobytes())
    h.update(synth.synth_lidar(ls, poses[6], "VLP64", 5, azimuth_steps=180).tobytes())
    h.update(synth.warm_guess(synth.rel_pose(poses[0], poses[1]), 3).tobytes())
    h.update(synth.perturb_pose(poses[7], np.random.default_rng(4242), sigma_t=(0.5, 0.5, 0.1), sigma_r_deg=(0.5, 0.5, 2.0)).tobytes())
    return h.hexdigest()


def test_inputs_digest_is_pinned():
    assert digest() == PINNED


@pytest.mark.gpu
def test_inputs_digest_is_pinned_on_the_gpu_box():
    assert digest() == PINNED


def test_own_elementary_functions_are_accurate():
    x = np.random.default_rng(0).uniform(-400.0, 400.0, 50000)
    assert np.abs(synth.dsin(x) - np.sin(x)).max() < 5e-16 and np.abs(synth.dcos(x) - np.cos(x)).max() < 5e-16
    u = np.random.default_rng(1).uniform(1e-300, 1.0, 50000)
    assert np.abs(synth.dlog(u) - np.log(u)).max() < 4e-15 * 700
    z = synth.dnormal(np.random.default_rng(2), 0.02, size=(400000,))
    assert abs(z.mean()) < 2e-4 and abs(z.std() - 0.02) < 1e-4 and abs(np.mean(z ** 4) / 0.02 ** 4 - 3.0) < 0.05
    R = synth.rot_xyz(0.3, -0.2, 1.1)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-15
    T = synth.make_pose([1.0, -2.0, 3.0], R)
    assert np.abs(synth.inv_pose(T) @ T - np.eye(4)).max() < 1e-15
    assert np.abs(synth.mat_mul(T, T) - T @ T).max() < 1e-14
