"""GPU: a scripted 64-keyframe ring session through the LoopDetector mirror — the batched GPU path (BatchMatcher: all candidates of a new
keyframe at once, candidates named by keyframe id, a second batch for the consistency check) against the reference's sequential loop run on
the CPU oracle (/root/reference/src/mrg_slam/loop_detector.cpp:97-303): the same list of loops, the same relative poses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("planar", [False, True])
def test_ring_session_gpu_batched_equals_sequential_oracle(planar):
    from loop_session import make_ring_session, run_session
    from mrg_slam_amd import BatchMatcher, NdtHip, prefilter, synth
    from mrg_slam_amd.loop_detector import LoopDetector
    from oracle import oracle as orc

    prm = {"use_planar_registration_guess": planar}
    reg_kw = dict(resolution=1.0, transformation_epsilon=0.01, maximum_iterations=64)
    sessions = {}
    for name in ("gpu_batched", "gpu_sequential", "oracle"):
        kfs, order = make_ring_session(64, "VLP64", prefilter=lambda c: prefilter(c, {"downsample_resolution": 0.2}))
        if name == "gpu_batched":
            det = LoopDetector(prm, matcher=BatchMatcher(**reg_kw))
        elif name == "gpu_sequential":
            det = LoopDetector(prm, registration=NdtHip(**reg_kw))
        else:
            det = LoopDetector(prm, registration=orc.Ndt(num_threads=8, **reg_kw))
        sessions[name] = (run_session(det, kfs, order), det, kfs)
    ref, det_o, kfs = sessions["oracle"]
    assert len(ref) >= 4 and any(lp.key1.slam_uuid != lp.key2.slam_uuid for lp in ref)  # loops on the second lap and between the robots
    assert np.mean([len(k.cloud) for k in kfs]) > 15000
    for name in ("gpu_batched", "gpu_sequential"):
        got, det, _ = sessions[name]
        assert [(lp.key1.id, lp.key2.id) for lp in got] == [(lp.key1.id, lp.key2.id) for lp in ref], name
        for a, b in zip(got, ref):
            assert np.linalg.norm(a.relative_pose[:3, 3].astype(np.float64) - b.relative_pose[:3, 3]) <= 1e-4, name
            assert synth.rotation_angle(a.relative_pose.astype(np.float64), b.relative_pose.astype(np.float64)) <= 1e-4, name
    # the batched path ran at most one alignment more per consistency check (the next keyframe beside the previous one)
    assert sessions["gpu_sequential"][1].alignments == det_o.alignments
    assert 0 <= sessions["gpu_batched"][1].alignments - det_o.alignments <= len(ref) + 8
    # the candidates' clouds stayed in the HBM keyframe store between calls
    assert sessions["gpu_batched"][1].matcher.store_bytes() > 0


def test_detect_batched_on_the_ring_session_equals_the_sequential_oracle():
    """LoopDetector::detect() receives SEVERAL new keyframes per call (loop_detector.cpp:18-21).  detect_batched aligns the candidates of all of them —
    the superset the LoopManager gates can only shrink — in ONE batch (one target grid per new keyframe), the consistency alignments in a second one, and
    replays gates, best-score rule and add_loop on the host.  Six keyframes per call on the 64-keyframe ring (5.4 m apart: a loop found for keyframe k
    prunes ALL same-robot candidates of k + 1 and k + 2 through the 15 m rule): the Loop list of the reference's sequential loop run on the CPU oracle
    with the same grouping, and of the one-keyframe-per-call session too wherever the grouping does not change the graph a call sees."""
    from loop_session import make_ring_session, run_session
    from mrg_slam_amd import BatchMatcher, prefilter, synth
    from mrg_slam_amd.loop_detector import LoopDetector
    from oracle import oracle as orc

    reg_kw = dict(resolution=1.0, transformation_epsilon=0.01, maximum_iterations=64)
    pf = lambda c: prefilter(c, {"downsample_resolution": 0.2})  # noqa: E731
    out = {}
    for name in ("batched", "oracle"):
        kfs, order = make_ring_session(64, "VLP64", prefilter=pf)
        det = LoopDetector(matcher=BatchMatcher(**reg_kw)) if name == "batched" else LoopDetector(registration=orc.Ndt(num_threads=8, **reg_kw))
        out[name] = (run_session(det, kfs, order, group=6, batched=(name == "batched")), det)
    got, det_b = out["batched"]
    ref, det_o = out["oracle"]
    assert len(ref) >= 3
    assert [(lp.key1.id, lp.key2.id) for lp in got] == [(lp.key1.id, lp.key2.id) for lp in ref]
    for a, b in zip(got, ref):
        assert np.linalg.norm(a.relative_pose[:3, 3].astype(np.float64) - b.relative_pose[:3, 3]) <= 1e-4
        assert synth.rotation_angle(a.relative_pose.astype(np.float64), b.relative_pose.astype(np.float64)) <= 1e-4
    # the gates did prune inside calls (the superset held more pairs than the sequential loop aligned), and the reference's counters agree
    assert det_b.alignments > det_o.alignments
    assert det_b.loop_candidates_sizes == det_o.loop_candidates_sizes
    assert det_b.average_time_per_candidate_us() < det_o.average_time_per_candidate_us()


def test_detect_batched_over_node_members_equals_one_batch():
    """The same detect_batched with a NodeMatcher as the matcher (mrgfe_node_*: the pair list of the call cut into contiguous blocks over the members — one per
    GPU on a real node, three sharing the card here): the superset batch and the consistency batch are sharded, the records gathered, and the Loop list with its
    relative poses is bit for bit the one-batch detector's."""
    from loop_session import make_ring_session, run_session
    from mrg_slam_amd import BatchMatcher, NodeMatcher, prefilter
    from mrg_slam_amd.loop_detector import LoopDetector

    reg_kw = dict(resolution=1.0, transformation_epsilon=0.01, maximum_iterations=64)
    pf = lambda c: prefilter(c, {"downsample_resolution": 0.2})  # noqa: E731
    out = {}
    for name in ("batch", "node"):
        kfs, order = make_ring_session(64, "VLP64", prefilter=pf)
        matcher = BatchMatcher(**reg_kw) if name == "batch" else NodeMatcher([0, 0, 0], **reg_kw)
        det = LoopDetector(matcher=matcher)
        out[name] = (run_session(det, kfs, order, group=6, batched=True), det)
    a, b = out["batch"][0], out["node"][0]
    assert len(a) >= 3 and [(lp.key1.id, lp.key2.id) for lp in a] == [(lp.key1.id, lp.key2.id) for lp in b]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.relative_pose, y.relative_pose)
    assert out["batch"][1].alignments == out["node"][1].alignments
