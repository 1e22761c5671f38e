"""GPU, run under the -DMRGFE_TESTING build of the library (MRGFE_LIB=mrg_slam_amd/libmrgfe_testing.so; started as ONE child process by
tests/test_gpu_hardening.py::test_fault_injection_suite_under_the_testing_library): the allocation-failure injector (mrgfe_dbg_fail_alloc_after) swept
over whole entry points — mrgfe_batch_align (NDT and GICP), mrgfe_prefilter, mrgfe_map_store_generate, mrgfe_node_align — shows every path unwinding
with an error code: no crash, no std::terminate from a joinable helper thread, and the next call (injector off) gives the right answer; a member made
to fail (mrgfe_dbg_node_fail_member) names itself and leaves the node usable; a failing align drains the zero-copy uploads it had queued."""
import ctypes as C

import numpy as np
import pytest

from oracle.replay import small_cloud

pytestmark = pytest.mark.gpu


def test_this_process_runs_the_testing_library():
    import os

    from mrg_slam_amd import _lib

    assert os.path.basename(_lib.LIB_PATH) == "libmrgfe_testing.so"
    assert hasattr(_lib.lib(), "mrgfe_dbg_fail_alloc_after") and hasattr(_lib.lib(), "mrgfe_dbg_node_fail_member")


def _sweep(make, run, check_ok, max_k=400):
    """fresh objects per k (grow-only workspaces would hide later allocations); every k must fail cleanly until one passes"""
    from mrg_slam_amd import MrgfeError
    from mrg_slam_amd._lib import lib

    failures = 0
    for k in range(max_k):
        obj = make()
        lib().mrgfe_dbg_fail_alloc_after(k)
        try:
            res = run(obj)
        except MrgfeError as e:
            assert "injected" in str(e) or "out of memory" in str(e).lower() or "member" in str(e), str(e)
            failures += 1
            continue
        finally:
            lib().mrgfe_dbg_fail_alloc_after(-1)
        check_ok(res)  # the injection point lay beyond the call's last allocation: the call ran to the end
        # a failed object must be reusable... the LAST failing one is gone; run once more on this one for the steady state
        check_ok(run(obj))
        return failures
    raise AssertionError(f"still failing after {max_k} injected allocation failures")


@pytest.mark.parametrize("method", ["NDT_HIP", "SMALL_GICP_HIP", "PCL_NDT_HIP"])
def test_allocation_failures_in_batch_align_unwind(method):
    from mrg_slam_amd import BatchMatcher, Context, _lib, synth
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc

    tgt = small_cloud(3000, 5)
    rng = np.random.default_rng(1)
    pairs = []
    for k in range(5):
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        pairs.append((orc.transform_points(np.linalg.inv(rel), tgt[: 1800 + 100 * k]), synth.perturb_pose(np.eye(4), rng)))
    prm = default_params(getattr(_lib, method))

    def run(bm):
        bm.clear()
        t = bm.add_target(tgt)
        for src, g in pairs:
            bm.add_pair(t, src, g)
        return bm.align(float("inf"))

    want = run(BatchMatcher(prm, Context(0)))

    def ok(res):
        assert res.tobytes() == want.tobytes()

    n = _sweep(lambda: BatchMatcher(prm, Context(0)), run, ok)
    assert n >= 10  # the sweep really walked through the call's allocations


def test_allocation_failures_in_prefilter_and_map_store_unwind():
    from mrg_slam_amd import Context, MapCloudStore, prefilter, synth

    raw = small_cloud(6000, 9, extent=(40, 30, 4))
    want = prefilter(raw)
    n = _sweep(lambda: Context(0), lambda ctx: prefilter(raw, ctx=ctx), lambda r: np.testing.assert_array_equal(r, want))
    assert n >= 5
    for mode in ({"outlier_removal_method": "STATISTICAL"}, {"downsample_method": "APPROX_VOXELGRID"}):
        w2 = prefilter(raw, mode)
        _sweep(lambda: Context(0), lambda ctx, mode=mode: prefilter(raw, mode, ctx=ctx), lambda r, w2=w2: np.testing.assert_array_equal(r, w2))
    clouds = [small_cloud(2000, 20 + k) for k in range(4)]
    poses = [synth.make_pose([2.0 * k, -1.0 * k, 0.0], synth.rot_xyz(0, 0, 0.2 * k)) for k in range(4)]
    ref_store = MapCloudStore(Context(0))
    for k, c in enumerate(clouds):
        ref_store.add(10 + k, c)
    want_map = ref_store.generate([10, 11, 12, 13], poses, resolution=0.25)

    def run(store):
        for k, c in enumerate(clouds):
            store.add(10 + k, c)  # (adding a key again with the same point count is a no-op)
        return store.generate([10, 11, 12, 13], poses, resolution=0.25)

    n = _sweep(lambda: MapCloudStore(Context(0)), run, lambda r: np.testing.assert_array_equal(r, want_map))
    assert n >= 5


def test_allocation_failures_in_node_align_name_the_member():
    from mrg_slam_amd import NodeMatcher, synth
    from mrg_slam_amd._lib import NDT_HIP
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc

    tgt = small_cloud(3000, 6)
    rng = np.random.default_rng(2)
    pairs = []
    for k in range(6):
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        pairs.append((orc.transform_points(np.linalg.inv(rel), tgt[: 1800 + 100 * k]), synth.perturb_pose(np.eye(4), rng)))
    prm = default_params(NDT_HIP)

    def run(node):
        node.clear()
        t = node.add_target(tgt)
        for src, g in pairs:
            node.add_pair(t, src, g)
        return node.align(float("inf"))

    want = run(NodeMatcher([0, 0], prm))
    n = _sweep(lambda: NodeMatcher([0, 0], prm), run, lambda r: r.tobytes() == want.tobytes() or (_ for _ in ()).throw(AssertionError("records differ")))
    assert n >= 10




def _workload(n_targets=3, n_pairs=11, seed=5, sizes=(5000, 3800, 4400)):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    targets = [small_cloud(sizes[k % len(sizes)], 300 + k) for k in range(n_targets)]
    rng = np.random.default_rng(seed)
    pairs = []
    for k in range(n_pairs):
        ti = min(n_targets - 1, k * n_targets // n_pairs)
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        src = orc.transform_points(np.linalg.inv(rel), targets[ti][: 2400 + 170 * k])
        pairs.append((ti, src, synth.perturb_pose(np.eye(4), rng)))
    return targets, pairs


def _one_batch(params, targets, pairs, fit=float("inf")):
    from mrg_slam_amd import BatchMatcher

    bm = BatchMatcher(params)
    tids = [bm.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        bm.add_pair(tids[ti], src, guess)
    res = bm.align(fit)
    res["pair_id"] = np.arange(len(pairs))
    return res


def _params(method, eps=0.01):
    from mrg_slam_amd.registration import default_params

    p = default_params(method)
    p.transformation_epsilon, p.maximum_iterations = eps, 64
    return p


def test_a_failing_member_returns_an_error_and_the_node_stays_usable():
    from mrg_slam_amd import MrgfeError, NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    targets, pairs = _workload(n_targets=2, n_pairs=6, seed=4)
    want = _one_batch(_params(NDT_HIP), targets, pairs)
    node = NodeMatcher([0, 0, 0], _params(NDT_HIP))

    def declare():
        node.clear()
        tids = [node.add_target(t) for t in targets]
        for ti, src, guess in pairs:
            node.add_pair(tids[ti], src, guess)

    declare()
    node.fail_member_once(1)
    with pytest.raises(MrgfeError, match=r"member 1 \(device 0\)"):
        node.align(float("inf"))
    declare()
    assert node.align(float("inf")).tobytes() == want.tobytes()
    # bad arguments are error codes, not crashes
    with pytest.raises(MrgfeError):
        node.add_pair(99, pairs[0][1], pairs[0][2])
    with pytest.raises(MrgfeError):
        NodeMatcher([], _params(NDT_HIP))
    with pytest.raises(MrgfeError):
        NodeMatcher([12345], _params(NDT_HIP))  # no such device
    # an empty list aligns to nothing
    node.clear()
    assert len(node.align()) == 0


def test_a_failing_align_drains_the_zero_copy_uploads_it_queued():
    """with mrgfe_ctx_set_zero_copy_uploads a cloud goes up by DMA out of the caller's page-locked buffer; mrgfe.h says the buffer is the
    caller's again when the consuming call returns — also when that call FAILS with the copy still queued.  An allocation failure inside the align: an
    error code comes back, the buffer is unpinned, overwritten and freed at once, and the next align (pageable copy of the same cloud) gives the records
    of an undisturbed run."""
    from mrg_slam_amd import BatchMatcher, Context, MrgfeError
    from mrg_slam_amd._lib import lib

    t = small_cloud(30000, seed=77)
    s = t[:29000].copy()
    ref = BatchMatcher(ctx=Context())
    ref.add_pair(ref.add_target(t), s, np.eye(4))
    want = ref.align().copy()
    pinned = t.copy()
    ctx = Context()  # (fresh: grow-only workspaces of the context above would leave the align nothing to allocate)
    ctx.set_zero_copy_uploads(True)
    assert lib().mrgfe_pin_host_buffer(ctx._h, pinned.ctypes.data_as(C.c_void_p), pinned.nbytes) == 0
    bm = BatchMatcher(ctx=ctx)
    bm.add_pair(bm.add_target(pinned), s, np.eye(4))
    lib().mrgfe_dbg_fail_alloc_after(2)
    try:
        with pytest.raises(MrgfeError):
            bm.align()
    finally:
        lib().mrgfe_dbg_fail_alloc_after(-1)
        assert lib().mrgfe_unpin_host_buffer(ctx._h, pinned.ctypes.data_as(C.c_void_p)) == 0
    pinned[:] = 0.0
    del pinned
    bm.clear()
    bm.add_pair(bm.add_target(t), s, np.eye(4))
    assert bm.align().tobytes() == want.tobytes()
