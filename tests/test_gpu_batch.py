"""GPU: batched candidate matching (LoopDetector::matching candidate loop) equals one-by-one registration."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def test_batch_equals_sequential_registrations():
    from mrg_slam_amd import BatchMatcher, NdtHip, synth
    from mrg_slam_amd.registration import result_matrix
    from oracle import oracle as orc

    targets = [small_cloud(5000, 100), small_cloud(3000, 101)]
    rng = np.random.default_rng(5)
    pairs = []
    for k in range(7):
        ti = k % 2
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        src = orc.transform_points(np.linalg.inv(rel), targets[ti][: 2500 + 300 * k])  # ragged sizes
        guess = synth.perturb_pose(np.eye(4), rng)
        pairs.append((ti, src, guess))
    bm = BatchMatcher(transformation_epsilon=0.01, maximum_iterations=64)
    tids = [bm.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        bm.add_pair(tids[ti], src, guess)
    res = bm.align(fitness_max_range=float("inf"))
    assert len(res) == len(pairs)
    for k, (ti, src, guess) in enumerate(pairs):
        reg = NdtHip(transformation_epsilon=0.01, maximum_iterations=64)
        reg.setInputTarget(targets[ti])
        reg.setInputSource(src)
        reg.align(guess)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation())
        assert res[k]["converged"] == int(reg.hasConverged())
        assert res[k]["iterations"] == reg.getFinalNumIteration()
        assert res[k]["evaluations"] == reg.evals
        assert res[k]["pair_id"] == k
        np.testing.assert_array_equal(res[k]["H"].reshape(6, 6), reg.getHessian())
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)
    ms, launches, nbytes = bm.kernel_stats()
    assert ms > 0 and launches > 0 and nbytes > 0
    # launched evaluations only: points x (16 + 7 x 8) + valid neighbours x 48 is the byte model of SURVEY.md §8(d)
    pts, nbrs = bm.pair_counts()
    assert pts > 0 and 0 < nbrs / pts <= 7
    assert nbytes == pytest.approx(pts * 72 + nbrs * 48, rel=1e-12)


def test_gicp_batch_equals_sequential_registrations_and_oracle():
    """GICP_HIP through the batch API (the loop-closure configuration of config/mrg_slam.yaml:181 uses a GICP back end):
    same results as one GicpHip object per pair and as the CPU oracle."""
    from mrg_slam_amd import BatchMatcher, GicpHip, synth
    from mrg_slam_amd._lib import GICP_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    targets = [small_cloud(4000, 200), small_cloud(3000, 201)]
    rng = np.random.default_rng(9)
    pairs = []
    for k in range(5):
        ti = k % 2
        rel = synth.make_pose(rng.normal(0, 0.15, 3), synth.rot_xyz(*rng.normal(0, 0.015, 3)))
        src = orc.transform_points(np.linalg.inv(rel), targets[ti][: 2000 + 250 * k])
        pairs.append((ti, src, synth.perturb_pose(np.eye(4), rng)))
    prm = default_params(GICP_HIP)
    prm.transformation_epsilon = 0.01
    bm = BatchMatcher(prm)
    tids = [bm.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        bm.add_pair(tids[ti], src, guess)
    res = bm.align(fitness_max_range=float("inf"))
    again = bm.align(fitness_max_range=float("inf"))  # cached target state: same answer
    for k, (ti, src, guess) in enumerate(pairs):
        reg = GicpHip(transformation_epsilon=0.01)
        reg.setInputTarget(targets[ti])
        reg.setInputSource(src)
        reg.align(guess)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation())
        np.testing.assert_array_equal(result_matrix(again[k]), reg.getFinalTransformation())
        assert res[k]["converged"] == int(reg.hasConverged()) and res[k]["iterations"] == reg.getFinalNumIteration()
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)
        o = orc.FastGicp(transformation_epsilon=0.01, num_threads=2)
        o.setInputTarget(targets[ti])
        o.setInputSource(src)
        o.align(guess)
        To = o.getFinalTransformation().astype(np.float64)
        Tg = result_matrix(res[k]).astype(np.float64)
        assert np.linalg.norm(Tg[:3, 3] - To[:3, 3]) < 1e-4 and synth.rotation_angle(Tg, To) < 1e-4
        assert res[k]["converged"] == int(o.hasConverged())


def test_batch_edge_cases():
    from mrg_slam_amd import BatchMatcher

    bm = BatchMatcher()
    assert len(bm.align()) == 0  # empty batch
    t = bm.add_target(small_cloud(2000, 1))
    e = bm.add_target(np.zeros((0, 4), np.float32))  # empty target: its pairs do not converge
    bm.add_pair(t, small_cloud(2000, 1))
    bm.add_pair(e, small_cloud(100, 2))
    bm.add_pair(t, np.zeros((0, 4), np.float32))  # empty source
    res = bm.align()
    assert res[0]["converged"] == 1
    assert res[1]["converged"] == 0 and res[2]["converged"] == 0


def test_two_contexts_from_two_threads_give_the_sequential_results():
    """Handles on distinct contexts (= HIP streams) are independent (SURVEY.md §8b threading row: the odometry and the
    loop-closure handles live in different threads of one process)."""
    import threading

    from mrg_slam_amd import BatchMatcher, Context, synth
    from mrg_slam_amd.registration import result_matrix
    from oracle import oracle as orc

    rng = np.random.default_rng(12)
    jobs = []
    for w in range(2):
        tgt = small_cloud(6000, 300 + w)
        pairs = []
        for k in range(6):
            rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
            pairs.append((orc.transform_points(np.linalg.inv(rel), tgt[: 3000 + 400 * k]), synth.perturb_pose(np.eye(4), rng)))
        jobs.append((tgt, pairs))

    def run(ctx, job, out, slot, repeats):
        tgt, pairs = job
        bm = BatchMatcher(ctx=ctx, transformation_epsilon=0.01)
        t = bm.add_target(tgt)
        for src, guess in pairs:
            bm.add_pair(t, src, guess)
        for _ in range(repeats):
            res = bm.align(fitness_max_range=float("inf"))
        out[slot] = res

    seq, par = [None, None], [None, None]
    ctxs = [Context(0), Context(0)]
    for w in range(2):
        run(ctxs[w], jobs[w], seq, w, 1)
    threads = [threading.Thread(target=run, args=(ctxs[w], jobs[w], par, w, 5)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for w in range(2):
        assert len(par[w]) == len(seq[w]) == 6
        for k in range(6):
            np.testing.assert_array_equal(result_matrix(par[w][k]), result_matrix(seq[w][k]))
            assert par[w][k]["fitness"] == seq[w][k]["fitness"] and par[w][k]["iterations"] == seq[w][k]["iterations"]


def test_context_confined_to_a_part_of_the_chip_gives_the_same_records():
    """mrgfe_ctx_create_reserving: the loop-closure context's streams (its own, the fitness side stream, the grid builders') carry a compute-unit
    mask; which compute units run a kernel enters no result (fixed-order sums everywhere), and a mask that leaves nothing is refused."""
    from mrg_slam_amd import BatchMatcher, Context, NdtHip, synth
    from mrg_slam_amd._lib import MrgfeError
    from mrg_slam_amd.registration import result_matrix
    from oracle import oracle as orc

    rng = np.random.default_rng(77)
    tgt = small_cloud(9000, 410)
    pairs = []
    for k in range(7):
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        pairs.append((orc.transform_points(np.linalg.inv(rel), tgt[: 4000 + 500 * k]), synth.perturb_pose(np.eye(4), rng)))

    def run(ctx):
        bm = BatchMatcher(ctx=ctx, transformation_epsilon=0.01)
        t = bm.add_target(tgt)
        for src, guess in pairs:
            bm.add_pair(t, src, guess)
        res = bm.align(fitness_max_range=float("inf"))
        reg = NdtHip(transformation_epsilon=0.01, ctx=ctx)
        reg.setInputTarget(tgt)
        reg.setInputSource(pairs[0][0])
        reg.align(pairs[0][1])
        return res, reg.getFinalTransformation().copy()

    full, full_single = run(Context(0))
    for reserve in (32, 200):
        part, part_single = run(Context(0, reserve_cus=reserve))
        for a, b in zip(full, part):
            np.testing.assert_array_equal(result_matrix(a), result_matrix(b))
            assert a["fitness"] == b["fitness"] and a["iterations"] == b["iterations"]
        np.testing.assert_array_equal(full_single, part_single)
    with pytest.raises(MrgfeError):
        Context(0, reserve_cus=100000)


@pytest.mark.parametrize("method", ["NDT_HIP", "GICP_HIP", "SMALL_GICP_HIP"])
def test_keyframe_store_gives_the_unkeyed_results(method):
    """Candidates added by keyframe id stay resident (cloud, GICP covariances) across clear(); a later batch that names
    them without handing the cloud over again gives exactly the results of plain add_pair calls."""
    from mrg_slam_amd import BatchMatcher, synth
    from mrg_slam_amd import _lib
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    prm = default_params(getattr(_lib, method))
    prm.transformation_epsilon = 0.01
    rng = np.random.default_rng(21)
    keyframes = {k: small_cloud(2500 + 200 * k, 300 + k) for k in (1, 2, 3)}
    plain, keyed = BatchMatcher(prm), BatchMatcher(prm)
    assert keyed.store_bytes() == 0 and keyed.has_cloud(1) is None
    for call in range(3):  # three "new keyframes", the same candidate pool every time
        tgt = small_cloud(3000, 400 + call)
        guesses = {k: synth.perturb_pose(np.eye(4), rng) for k in keyframes}
        plain.clear()
        keyed.clear()
        tp, tk = plain.add_target(tgt), keyed.add_target(tgt)
        for k, cloud in keyframes.items():
            plain.add_pair(tp, cloud, guesses[k])
            keyed.add_pair(tk, cloud if call == 0 else None, guesses[k], key=k)  # later calls: by id only
        if call == 2:  # the same keyframe against the target twice in one batch
            plain.add_pair(tp, keyframes[2], np.eye(4))
            keyed.add_pair(tk, None, np.eye(4), key=2)
        a, b = plain.align(float("inf")), keyed.align(float("inf"))
        for i in range(len(a)):
            np.testing.assert_array_equal(result_matrix(a[i]), result_matrix(b[i]))
            assert a[i]["converged"] == b[i]["converged"] and a[i]["iterations"] == b[i]["iterations"]
            assert a[i]["fitness"] == b[i]["fitness"]
    per_point = 16 if method == "NDT_HIP" else 64
    assert keyed.store_bytes() >= per_point * sum(len(c) for c in keyframes.values())
    assert keyed.has_cloud(2) == len(keyframes[2])
    with pytest.raises(KeyError):
        keyed.add_pair(0, None, np.eye(4), key=99)
    # a key can be re-bound to another cloud once no batch uses it; forgetting frees the memory
    keyed.clear()
    t = keyed.add_target(small_cloud(2000, 1))
    keyed.add_pair(t, small_cloud(1000, 2), np.eye(4), key=2)
    assert keyed.has_cloud(2) == 1000
    with pytest.raises(RuntimeError):
        keyed.forget(2)  # in use
    keyed.clear()
    keyed.forget()
    assert keyed.store_bytes() == 0


def test_keyframe_store_evicts_least_recently_used(monkeypatch):
    from mrg_slam_amd import BatchMatcher

    monkeypatch.setenv("MRGFE_KEYFRAME_STORE_MB", "1")  # 1 MiB: room for about three 20k-point clouds of 320 KB
    bm = BatchMatcher()
    for call, key in enumerate((1, 2, 3, 4, 5)):
        bm.clear()
        t = bm.add_target(small_cloud(1000, 5))
        bm.add_pair(t, small_cloud(20000, 50 + key), np.eye(4), key=key)
        if call >= 1:
            bm.add_pair(t, None, np.eye(4), key=1)  # keyframe 1 is used every time: it must survive
    assert bm.has_cloud(1) == 20000 and bm.has_cloud(5) == 20000
    assert bm.has_cloud(2) is None  # oldest unused one went first
    assert bm.store_bytes() <= (1 << 20) + 20000 * 16


def test_device_resident_clouds_give_the_host_pointer_results():
    """The *_device entry points reference clouds already in HBM (packed float4, e.g. torch tensors): same results as
    handing the same clouds over as host pointers, for the batch and for a single registration."""
    import torch

    from mrg_slam_amd import BatchMatcher, NdtHip, synth
    from mrg_slam_amd.registration import result_matrix

    rng = np.random.default_rng(33)
    tgt = small_cloud(4000, 500)
    srcs = [small_cloud(2500 + 100 * k, 510 + k) for k in range(3)]
    guesses = [synth.perturb_pose(np.eye(4), rng) for _ in srcs]
    d_tgt = torch.from_numpy(tgt).cuda(0)
    d_srcs = [torch.from_numpy(s).cuda(0) for s in srcs]
    torch.cuda.synchronize()
    host, dev = BatchMatcher(transformation_epsilon=0.01), BatchMatcher(transformation_epsilon=0.01)
    th, td = host.add_target(tgt), dev.add_target_device(d_tgt.data_ptr(), len(tgt))
    for s, ds, g in zip(srcs, d_srcs, guesses):
        host.add_pair(th, s, g)
        dev.add_pair_device(td, ds.data_ptr(), len(s), g)
    a, b = host.align(float("inf")), dev.align(float("inf"))
    for i in range(len(srcs)):
        np.testing.assert_array_equal(result_matrix(a[i]), result_matrix(b[i]))
        assert a[i]["fitness"] == b[i]["fitness"] and a[i]["iterations"] == b[i]["iterations"]
    r1, r2 = NdtHip(transformation_epsilon=0.01), NdtHip(transformation_epsilon=0.01)
    r1.setInputTarget(tgt)
    r1.setInputSource(srcs[0])
    r2.setInputTargetDevice(d_tgt.data_ptr(), len(tgt))
    r2.setInputSourceDevice(d_srcs[0].data_ptr(), len(srcs[0]))
    r1.align(guesses[0])
    r2.align(guesses[0])
    np.testing.assert_array_equal(r1.getFinalTransformation(), r2.getFinalTransformation())
    assert r1.getFitnessScore() == r2.getFitnessScore()


def test_fitness_grids_wait_for_the_target_upload(monkeypatch):
    """The steady state of INTEGRATION.md: ONE new keyframe handed over as host records (asynchronous copy + device gather on the
    batch's stream), every candidate resident in the keyframe store, then align.  The fitness grids are built on helper streams
    beside the alignment: they must see the finished upload (round-2 advisor finding) — same scores as building them afterwards
    on the batch's own stream, call after call."""
    from mrg_slam_amd import BatchMatcher
    from mrg_slam_amd._lib import LAYOUT_PCL_XYZI

    rng = np.random.default_rng(77)
    cands = {k: small_cloud(20000, 900 + k) for k in (1, 2, 3, 4)}
    bm = BatchMatcher(transformation_epsilon=0.1)
    t0 = bm.add_target(small_cloud(1000, 1))
    for k, c in cands.items():
        bm.add_pair(t0, c, np.eye(4), key=k)  # fills the store
    bm.align(float("inf"))
    for call in range(6):
        tgt = small_cloud(200000, 950 + call)  # large: the copy and the gather are still in flight when the builders start
        rec = np.zeros((len(tgt), 8), np.float32)
        rec[:, :3] = tgt[:, :3]
        rec[:, 3] = 1.0
        rec[:, 4] = tgt[:, 3]
        got = []
        for no_overlap in (False, True):
            if no_overlap:
                monkeypatch.setenv("MRGFE_NO_FIT_OVERLAP", "1")
            else:
                monkeypatch.delenv("MRGFE_NO_FIT_OVERLAP", raising=False)
            bm.clear()
            t = bm.add_target_records(rec, len(tgt), LAYOUT_PCL_XYZI)
            for k in cands:
                bm.add_pair(t, None, np.eye(4), key=k)
            got.append(bm.align(float("inf")))
        for a, b in zip(*got):
            assert a["fitness"] == b["fitness"] and np.isfinite(a["fitness"])
            np.testing.assert_array_equal(a["T"], b["T"])


def test_batch_timing_is_the_references_per_candidate_statistic():
    """mrgfe_batch_timing: average_time_per_candidate_us as apps/mrg_slam_component.cpp:1032-1037 writes it (total loop-detection wall time over total
    candidates, loop_detector.cpp:22-34) for a batch object: queueing (from the clear) to records, over the pairs aligned; totals add up over calls."""
    import time

    from mrg_slam_amd import BatchMatcher

    tgt = small_cloud(6000, 3)
    bm = BatchMatcher(transformation_epsilon=0.01, maximum_iterations=64)
    assert bm.timing() == {"average_time_per_candidate_us": 0.0, "last_align_us": 0.0, "last_align_pairs": 0, "total_pairs": 0}
    walls = []
    for n in (3, 5):
        t0 = time.perf_counter()
        bm.clear()
        t = bm.add_target(tgt)
        for k in range(n):
            bm.add_pair(t, tgt[: 4000 + 100 * k].copy(), np.eye(4))
        bm.align(float("inf"))
        walls.append(1e6 * (time.perf_counter() - t0))
        tm = bm.timing()
        assert tm["last_align_pairs"] == n and 0 < tm["last_align_us"] <= walls[-1]
        assert tm["last_align_us"] > 0.5 * walls[-1]  # the statistic covers the call sequence, not just a kernel
    tm = bm.timing(reset=True)
    assert tm["total_pairs"] == 8 and abs(tm["average_time_per_candidate_us"] * 8 - sum(walls)) < 0.5 * sum(walls)
    assert bm.timing()["total_pairs"] == 0
