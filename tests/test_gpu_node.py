"""GPU: mrgfe_node_* — the loop-closure candidate batch over several members (one per GPU on a real node; here N members on the one card of the
box, which is what the header allows for exactly this purpose): the gathered records must equal those of ONE batch holding the whole pair list bit
for bit, for uneven blocks, keyed and unkeyed clouds, every method the batch API serves (a failing member: tests/faultinject/); the RCCL gather is exercised with one member (RCCL refuses duplicate devices)."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _workload(n_targets=3, n_pairs=11, seed=5, sizes=(5000, 3800, 4400)):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    targets = [small_cloud(sizes[k % len(sizes)], 300 + k) for k in range(n_targets)]
    rng = np.random.default_rng(seed)
    pairs = []
    for k in range(n_pairs):
        ti = min(n_targets - 1, k * n_targets // n_pairs)  # ordered by target, like the reference's list (new keyframe after new keyframe)
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


@pytest.mark.parametrize("members", [1, 2, 3, 4])
def test_node_records_equal_one_batch(members):
    from mrg_slam_amd import NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    targets, pairs = _workload()
    want = _one_batch(_params(NDT_HIP), targets, pairs)
    node = NodeMatcher([0] * members, _params(NDT_HIP))
    assert node.n_members == members
    for rep in range(2):  # the second call reuses the members' workspaces
        node.clear()
        tids = [node.add_target(t) for t in targets]
        for ti, src, guess in pairs:
            node.add_pair(tids[ti], src, guess)
        got = node.align(float("inf"))
        assert got.tobytes() == want.tobytes()
    blocks = [node.shard(m) for m in range(members)]
    assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == len(pairs) and max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1
    assert all(blocks[m][0] + blocks[m][1] == blocks[m + 1][0] for m in range(members - 1))
    assert node.last_gather() == "host"  # members share a card: RCCL refuses duplicate devices


@pytest.mark.parametrize("method", ["GICP_HIP", "SMALL_GICP_HIP", "VGICP_HIP", "PCL_NDT_HIP"])
def test_node_serves_every_batch_method(method):
    from mrg_slam_amd import NodeMatcher, _lib

    targets, pairs = _workload(n_targets=2, n_pairs=5, seed=8)
    prm = _params(getattr(_lib, method), eps=1e-4 if method == "PCL_NDT_HIP" else 0.01)
    want = _one_batch(prm, targets, pairs)
    node = NodeMatcher([0, 0], prm)
    tids = [node.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        node.add_pair(tids[ti], src, guess)
    assert node.align(float("inf")).tobytes() == want.tobytes()


def test_keyed_clouds_stay_resident_and_may_be_named_without_data():
    from mrg_slam_amd import NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    targets, pairs = _workload(n_targets=2, n_pairs=7, seed=3)
    want = _one_batch(_params(NDT_HIP), targets, pairs)
    node = NodeMatcher([0, 0, 0], _params(NDT_HIP))
    for rep in range(3):
        node.clear()
        tids = [node.add_target(t if rep < 2 else None, key=1000 + k, n_points=len(t)) for k, t in enumerate(targets)]
        for k, (ti, src, guess) in enumerate(pairs):
            # first call: clouds handed over; later calls: the same list names them by key only (same blocks -> same members hold them)
            node.add_pair(tids[ti], src if rep == 0 else None, guess, key=50 + k, n_points=len(src))
        got = node.align(float("inf"))
        assert got.tobytes() == want.tobytes(), rep
    assert node.store_bytes() >= sum(len(p[1]) for p in pairs) * 16
    node.forget(0)
    assert node.store_bytes() == 0
    node.clear()
    t = node.add_target(None, key=1000, n_points=len(targets[0]))  # forgotten: naming it without data is now an error of the member that needs it
    node.add_pair(t, pairs[0][1], pairs[0][2])
    from mrg_slam_amd import MrgfeError

    with pytest.raises(MrgfeError, match="member 0"):
        node.align()


def test_member_target_store_is_bounded_lru(monkeypatch):
    """in the LoopDetector flow every keyframe is a target exactly once (loop_detector.cpp:104), so the members' keyed TARGET store must not
    grow by a cloud per keyframe for the life of the process.  With a small cap (MRGFE_KEYFRAME_STORE_MB, the batch store's knob) the least recently
    named targets are dropped — never one the running call names — and naming a dropped key without data is the usual error."""
    from mrg_slam_amd import MrgfeError, NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    monkeypatch.setenv("MRGFE_KEYFRAME_STORE_MB", "1")  # 1 MiB: three ~20k-point targets (0.3 MB each) fit, the fourth evicts the oldest
    node = NodeMatcher([0], _params(NDT_HIP))
    monkeypatch.delenv("MRGFE_KEYFRAME_STORE_MB")
    targets = [small_cloud(20000, 900 + k) for k in range(6)]
    src = targets[0][:3000].copy()
    held = []
    for k, t in enumerate(targets):  # one new keyframe per call, like matching()
        node.clear()
        node.add_pair(node.add_target(t, key=7000 + k, n_points=len(t)), src, np.eye(4))
        node.align()
        held.append(node.store_bytes())
    assert max(held) <= (1 << 20) + 20000 * 16 and held[-1] <= (1 << 20)
    # the most recent keys are still resident and can be named without data; the oldest is gone
    node.clear()
    node.add_pair(node.add_target(None, key=7005, n_points=len(targets[5])), src, np.eye(4))
    node.add_pair(node.add_target(None, key=7004, n_points=len(targets[4])), src, np.eye(4))
    assert len(node.align()) == 2
    node.clear()
    node.add_pair(node.add_target(None, key=7000, n_points=len(targets[0])), src, np.eye(4))
    with pytest.raises(MrgfeError, match="not resident"):
        node.align()
    # a call whose OWN targets exceed the cap keeps all of them (nothing it names is evicted) and gives the records of an unbounded node
    big = NodeMatcher([0], _params(NDT_HIP))
    for nd in (node, big):
        nd.clear()
        for k, t in enumerate(targets):
            nd.add_pair(nd.add_target(t, key=8000 + k, n_points=len(t)), targets[k][:2500 + 100 * k].copy(), np.eye(4))
    assert node.align(float("inf")).tobytes() == big.align(float("inf")).tobytes()


def test_node_bad_arguments_are_error_codes():
    from mrg_slam_amd import MrgfeError, NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    targets, pairs = _workload(n_targets=2, n_pairs=6, seed=4)
    node = NodeMatcher([0, 0, 0], _params(NDT_HIP))
    with pytest.raises(MrgfeError):
        node.add_pair(99, pairs[0][1], pairs[0][2])
    with pytest.raises(MrgfeError):
        NodeMatcher([], _params(NDT_HIP))
    with pytest.raises(MrgfeError):
        NodeMatcher([12345], _params(NDT_HIP))  # no such device
    # an empty list aligns to nothing
    node.clear()
    assert len(node.align()) == 0


def test_select_best_is_the_references_sequential_rule():
    from mrg_slam_amd import NodeMatcher, loop_closure
    from mrg_slam_amd.registration import RESULT_DTYPE

    rng = np.random.default_rng(0)
    rec = np.zeros(40, dtype=RESULT_DTYPE)
    rec["fitness"] = rng.choice([0.1, 0.2, 0.2, 0.5, np.finfo(np.float64).max, np.inf, np.nan], 40)
    rec["converged"] = rng.random(40) < 0.7
    group_first = [0, 7, 7, 19, 40]  # an empty group among them
    got = NodeMatcher.select_best(rec, group_first)
    for g in range(4):
        assert got[g] == loop_closure.select_best(rec[group_first[g]:group_first[g + 1]]) or (got[g][0] is None and loop_closure.select_best(rec[group_first[g]:group_first[g + 1]])[0] is None)


def test_rccl_gather_with_one_member(monkeypatch):
    """the RCCL path (dlopen, ncclCommInitAll, grouped ncclAllGather, unpacking by pair id) on the one device of the box"""
    from mrg_slam_amd import NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP

    monkeypatch.setenv("MRGFE_NODE_GATHER", "rccl")
    targets, pairs = _workload(n_targets=2, n_pairs=5, seed=6)
    want = _one_batch(_params(NDT_HIP), targets, pairs)
    node = NodeMatcher([0], _params(NDT_HIP))
    tids = [node.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        node.add_pair(tids[ti], src, guess)
    got = node.align(float("inf"))
    assert node.last_gather() == "rccl"
    assert got.tobytes() == want.tobytes()


def test_node_reads_page_locked_clouds_by_dma_with_the_same_records():
    """The members of a node run with zero-copy uploads on (mrgfe.h: the node's clouds are declared by pointer and uploaded inside mrgfe_node_align):
    clouds of 64 KB and more in page-locked memory go up by DMA from the caller's slab, smaller and pageable ones through the staging ring — keyed
    (resident) targets included.  Same records as one batch fed pageable copies."""
    import ctypes as C

    from mrg_slam_amd import Context, NodeMatcher
    from mrg_slam_amd._lib import NDT_HIP, lib

    targets, pairs = _workload(n_targets=3, n_pairs=9, sizes=(9000, 5200, 4400))
    want = _one_batch(_params(NDT_HIP), targets, pairs)
    ctx = Context()
    total = sum(len(t) for t in targets) + sum(len(p[1]) for p in pairs)
    slab = np.empty((total, 4), np.float32)
    assert lib().mrgfe_pin_host_buffer(ctx._h, slab.ctypes.data_as(C.c_void_p), slab.nbytes) == 0
    try:
        o = 0

        def put(a):
            nonlocal o
            v = slab[o:o + len(a)]
            v[:] = a
            o += len(a)
            return v

        p_targets = [put(t) for t in targets]
        p_pairs = [(ti, put(src), g) for ti, src, g in pairs]
        node = NodeMatcher([0, 0], _params(NDT_HIP))
        for rep in range(2):  # (the second call finds the keyed target resident)
            node.clear()
            tids = [node.add_target(t, key=(7 if k == 0 else 0)) for k, t in enumerate(p_targets)]
            for ti, src, guess in p_pairs:
                node.add_pair(tids[ti], src, guess)
            got = node.align(float("inf"))
            assert got.tobytes() == want.tobytes()
    finally:
        assert lib().mrgfe_unpin_host_buffer(ctx._h, slab.ctypes.data_as(C.c_void_p)) == 0
