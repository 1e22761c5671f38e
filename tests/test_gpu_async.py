"""GPU: mrgfe_batch_align_async / mrgfe_batch_wait — a batch's align on a worker thread of its own, so that a single-threaded caller keeps two
batches (two contexts) in flight.  Records equal the synchronous call's bit for bit; misuse returns error codes."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _workload(seed, n_pairs=9):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt = small_cloud(5000, 400 + seed)
    rng = np.random.default_rng(seed)
    pairs = []
    for k in range(n_pairs):
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        pairs.append((orc.transform_points(np.linalg.inv(rel), tgt[: 2600 + 150 * k]), synth.perturb_pose(np.eye(4), rng)))
    return tgt, pairs


def _queue(bm, tgt, pairs):
    bm.clear()
    t = bm.add_target(tgt)
    for src, guess in pairs:
        bm.add_pair(t, src, guess)


@pytest.mark.parametrize("method", ["NDT_HIP", "SMALL_GICP_HIP"])
def test_two_batches_in_flight_give_the_synchronous_records(method):
    from mrg_slam_amd import BatchMatcher, Context, _lib
    from mrg_slam_amd.registration import default_params

    prm = default_params(getattr(_lib, method))
    loads = [_workload(s) for s in range(4)]
    ref = BatchMatcher(prm)
    want = []
    for tgt, pairs in loads:
        _queue(ref, tgt, pairs)
        want.append(ref.align(float("inf")))
    bms = [BatchMatcher(prm, Context(0)), BatchMatcher(prm, Context(0))]
    got = [None] * len(loads)
    for rep in range(2):
        for k, (tgt, pairs) in enumerate(loads):
            b = bms[k % 2]
            if k >= 2:
                got[k - 2] = b.wait()
            _queue(b, tgt, pairs)  # (blocks while this batch's previous align is still running: the worker holds its context)
            b.align_async(float("inf"))
        for k in (len(loads) - 2, len(loads) - 1):
            got[k] = bms[k % 2].wait()
        for k in range(len(loads)):
            assert got[k].tobytes() == want[k].tobytes(), (rep, k)


def test_misuse_returns_error_codes():
    from mrg_slam_amd import BatchMatcher, MrgfeError
    from mrg_slam_amd._lib import check, lib

    tgt, pairs = _workload(7, 4)
    bm = BatchMatcher()
    with pytest.raises(MrgfeError):
        check(lib().mrgfe_batch_wait(bm._h))  # nothing was started
    _queue(bm, tgt, pairs)
    bm.align_async()
    with pytest.raises(MrgfeError, match="call mrgfe_batch_wait first"):
        bm.align_async()  # one align in flight per batch (running, or finished and not yet waited for)
    res = bm.wait()
    assert len(res) == 4 and res["converged"].all()
    with pytest.raises(MrgfeError):
        check(lib().mrgfe_batch_wait(bm._h))
    assert lib().mrgfe_batch_align_async(None, -1.0, None) < 0
    # a batch destroyed with an align in flight waits for it
    _queue(bm, tgt, pairs)
    bm.align_async()
    del bm


def test_zero_copy_uploads_with_two_batches_in_flight():
    """mrgfe_ctx_set_zero_copy_uploads + mrgfe_batch_align_async: page-locked clouds are read by DMA while the OTHER batch aligns; the clouds stay
    unchanged until the wait, as the switch's contract asks.  Records equal the synchronous ones from pageable copies."""
    import ctypes as C

    from mrg_slam_amd import BatchMatcher, Context
    from mrg_slam_amd._lib import lib

    rng = np.random.default_rng(11)
    clouds = [small_cloud(20000 + 500 * k, 900 + k) for k in range(4)]
    guesses = []
    for _ in range(3):
        g = np.eye(4)
        g[:3, 3] = rng.uniform(-0.2, 0.2, 3)
        guesses.append(g)

    def fill(bm, cs):
        bm.clear()
        for k in range(3):
            bm.add_pair(bm.add_target(cs[k]), cs[k + 1][: len(cs[k + 1]) - 100], guesses[k])

    ref = BatchMatcher(ctx=Context())
    fill(ref, clouds)
    want = ref.align(float("inf"))
    ctxs = [Context(), Context()]
    slab = np.empty((sum(len(c) for c in clouds), 4), np.float32)
    assert lib().mrgfe_pin_host_buffer(ctxs[0]._h, slab.ctypes.data_as(C.c_void_p), slab.nbytes) == 0
    try:
        o, pinned = 0, []
        for c in clouds:
            v = slab[o:o + len(c)]
            v[:] = c
            o += len(c)
            pinned.append(v)
        bms = []
        for c in ctxs:
            c.set_zero_copy_uploads(True)
            bms.append(BatchMatcher(ctx=c))
        for rep in range(3):
            for bm in bms:
                fill(bm, pinned)
                bm.align_async(float("inf"))
            for bm in bms:
                assert bm.wait().tobytes() == want.tobytes()
    finally:
        assert lib().mrgfe_unpin_host_buffer(ctxs[0]._h, slab.ctypes.data_as(C.c_void_p)) == 0
