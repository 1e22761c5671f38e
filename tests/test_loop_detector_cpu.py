"""CPU: the LoopDetector mirror (mrg_slam_amd/loop_detector.py) — candidate selection, best-score rule, fitness gate, consistency check,
loop manager — and that its batched path takes the decisions of the reference's sequential loop
(/root/reference/src/mrg_slam/loop_detector.cpp:41-303).  The registrations here are scripted; the GPU suite runs real ones."""
import numpy as np
import pytest

from mrg_slam_amd import synth
from mrg_slam_amd.loop_detector import DEFAULTS, Edge, KeyFrame, LoopDetector, angular_distance_to_identity, normalize_estimate


def _kf(i, x, y, accum, uuid="a", first=False, yaw=0.0):
    cloud = np.full((1, 4), i, dtype=np.float32)  # the scripted registrations read the keyframe id out of its "cloud"
    return KeyFrame(id=i, cloud=cloud, estimate=synth.make_pose([x, y, 0.0], synth.rot_z(yaw)), accum_distance=accum, slam_uuid=uuid, first_keyframe=first)


class ScriptedRegistration:
    """pcl::Registration call surface; results come from a table keyed by (target id, source id)."""

    def __init__(self, table):
        self.table, self.calls = table, []

    def setInputTarget(self, cloud):
        self.t = int(cloud[0, 0])

    def setInputSource(self, cloud):
        self.s = int(cloud[0, 0])

    def align(self, guess):
        self.calls.append((self.t, self.s))
        self.guess = np.asarray(guess)

    def _rec(self):
        return self.table[(self.t, self.s)]

    def hasConverged(self):
        return self._rec()[1]

    def getFinalTransformation(self):
        T = self._rec()[0]
        return (self.guess if T is None else T).astype(np.float32)

    def getFitnessScore(self, max_range):
        return self._rec()[2]


class ScriptedMatcher:
    """BatchMatcher call surface over the same table."""

    def __init__(self, table):
        self.table, self.store, self.ids, self.calls = table, {}, {}, []

    def clear(self):
        self.pairs, self.targets = [], []

    def add_target(self, cloud):
        self.targets.append(int(cloud[0, 0]))
        return len(self.targets) - 1

    def has_cloud(self, key):
        return self.store.get(key)

    def forget(self, key=0):
        self.store.pop(key, None)
        self.ids.pop(key, None)

    def add_pair(self, t, cloud, guess, key=0):
        if cloud is None:
            assert key in self.store
        else:
            self.store[key] = len(cloud)
            self.ids[key] = int(cloud[0, 0])  # the store key names (slam_uuid, id, content); the table is keyed by keyframe id
        self.pairs.append((self.targets[t], self.ids[key], np.asarray(guess)))

    def align(self, fitness_max_range):
        from mrg_slam_amd.registration import RESULT_DTYPE

        out = np.zeros(len(self.pairs), dtype=RESULT_DTYPE)
        for i, (t, key, guess) in enumerate(self.pairs):
            T, conv, score = self.table[(t, key)]
            self.calls.append((t, key))
            out[i]["T"] = (guess if T is None else T).astype(np.float32).T.reshape(-1)
            out[i]["converged"] = conv
            out[i]["fitness"] = score if fitness_max_range >= 0 else np.finfo(np.float64).max
        return out


def test_find_candidates_follows_the_reference_rules():
    det = LoopDetector(registration=ScriptedRegistration({}))
    new = _kf(100, 0.0, 0.0, 100.0)
    kfs = [_kf(1, 3.0, 0.0, 0.0, first=True),        # first keyframe of a robot: never a candidate
           _kf(2, 3.0, 4.0, 10.0),                    # 5 m away, 90 m of driving in between: a candidate
           _kf(3, 30.0, 0.0, 20.0),                   # too far in xy
           _kf(4, 1.0, 1.0, 90.0),                    # same robot, only 10 m of driving since: too recent
           _kf(5, 2.0, 2.0, 3.0, uuid="b"),           # other robot: no accumulated-distance rule without an earlier loop
           _kf(6, 0.0, 14.9, 50.0)]                   # just inside candidate_max_xy_distance
    new.connected.add(6)                              # there is already an edge
    assert [c.id for c in det.find_candidates(new, kfs)] == [2, 5]
    # a recent loop of the same SLAM instance pair blocks new candidates until enough distance has accumulated
    from mrg_slam_amd.loop_detector import Loop

    det.loop_manager.add_loop(Loop(_kf(90, 0, 0, 92.0), kfs[1], np.eye(4, dtype=np.float32)))
    assert [c.id for c in det.find_candidates(new, kfs)] == [5]
    det.loop_manager.add_loop(Loop(_kf(91, 0, 0, 97.0), kfs[4], np.eye(4, dtype=np.float32)))  # other robot: 5 m rule, 3 m driven
    assert det.find_candidates(new, kfs) == []


def _chain(n):
    kfs = [_kf(i + 1, 2.0 * i, 0.0, 30.0 * i, first=(i == 0)) for i in range(n)]
    for a, b in zip(kfs[1:], kfs[:-1]):
        rel = np.linalg.inv(a.estimate) @ b.estimate
        a.prev_edge = Edge(a, b, rel)
        b.next_edge = Edge(a, b, rel)
    return kfs


@pytest.mark.parametrize("planar", [False, True])
def test_batched_matching_takes_the_sequential_decisions(planar):
    """Best-score rule with equal scores and a non-converged candidate, the fitness gate, the consistency check through the previous and,
    when that fails, the next keyframe: both paths return the same loops and run the alignments the reference would."""
    kfs = _chain(6)
    new = _kf(50, 5.0, 1.0, 500.0)
    new.estimate[2, 3] = 0.4
    good = lambda kf, dx=0.0: ((np.linalg.inv(normalize_estimate(new.estimate)) @ kf.estimate @ synth.make_pose([dx, 0, 0], np.eye(3))).astype(np.float32))  # noqa: E731
    cases = {
        "prev consistent": {(50, 2): (good(kfs[1]), True, 0.5), (50, 3): (good(kfs[2]), True, 0.4), (50, 4): (good(kfs[3]), True, 0.4), (50, 5): (good(kfs[4]), False, 0.1), (50, 6): (good(kfs[5]), True, 0.9)},
        "next rescues": {(50, 2): (good(kfs[1], 2.0), True, 0.5), (50, 3): (good(kfs[2]), True, 0.3), (50, 4): (good(kfs[3]), True, 0.35), (50, 5): (good(kfs[4]), True, 2.0), (50, 6): (good(kfs[5]), True, 2.0)},
        "inconsistent": {(50, 2): (good(kfs[1], 2.0), True, 0.5), (50, 3): (good(kfs[2]), True, 0.3), (50, 4): (good(kfs[3], -2.0), True, 0.35), (50, 5): (good(kfs[4]), True, 2.0), (50, 6): (good(kfs[5]), True, 2.0)},
        "over the fitness threshold": {(50, k): (good(kfs[k - 1]), True, 1.3 + 0.1 * k) for k in range(2, 7)},
        "nothing converged": {(50, k): (good(kfs[k - 1]), False, 0.1) for k in range(2, 7)},
    }
    expect = {"prev consistent": 4, "next rescues": 3, "inconsistent": None, "over the fitness threshold": None, "nothing converged": None}
    for name, table in cases.items():
        seq = LoopDetector({"use_planar_registration_guess": planar}, registration=ScriptedRegistration(table))
        bat = LoopDetector({"use_planar_registration_guess": planar}, matcher=ScriptedMatcher(table))
        cands = kfs[1:]
        ls, lb = seq.matching(cands, new), bat.matching(cands, new)
        assert (ls is None) == (lb is None) == (expect[name] is None), name
        if ls is not None:
            assert ls.key2.id == lb.key2.id == expect[name], name  # "prev consistent": the LAST of the two equal best scores wins
            np.testing.assert_array_equal(ls.relative_pose, lb.relative_pose)
            assert seq.loop_manager.get_loop("a", "a") is ls
        # the batched path aligns the next keyframe together with the previous one; the sequential one only when the previous one failed
        assert bat.alignments - seq.alignments in (0, 1), name
        if planar:
            assert seq.registration.guess[2, 3] == 0.0


def test_keyframe_store_is_used_by_the_batched_path():
    kfs = _chain(4)
    table = {(t, s): (None, True, 0.2) for t in (50, 51) for s in range(1, 5)}
    m = ScriptedMatcher(table)
    det = LoopDetector({"enable_loop_closure_consistency_check": False}, matcher=m)
    det.matching(kfs[1:], _kf(50, 1.0, 0.0, 400.0))
    assert m.store == {kf.store_key(): 1 for kf in kfs[1:]} and sorted(m.ids.values()) == [2, 3, 4]
    det.matching(kfs[1:], _kf(51, 1.0, 0.0, 500.0))  # the second new keyframe names the candidates by id only (add_pair asserts it)


def test_normalize_estimate_and_angular_distance():
    R = synth.rot_xyz(0.3, -0.2, 1.1)
    T = synth.make_pose([1, 2, 3], R * 1.0001)  # slightly denormalised, like an optimiser's output
    N = normalize_estimate(T)
    np.testing.assert_allclose(N[:3, :3] @ N[:3, :3].T, np.eye(3), atol=1e-12)
    np.testing.assert_allclose(N[:3, 3], [1, 2, 3])
    assert angular_distance_to_identity(R) == pytest.approx(synth.rotation_angle(R, np.eye(3)), abs=1e-6)
    assert angular_distance_to_identity(np.eye(3)) == 0.0
    assert DEFAULTS["fitness_score_thresh"] == 1.25 and DEFAULTS["loop_closure_consistency_max_delta_angle"] == pytest.approx(np.deg2rad(3.0), abs=1e-6)


def test_scripted_ring_session_on_the_oracle():
    """A small session (VLP-16, 28 keyframes) through the sequential path with the CPU oracle as the registration: loops are found where
    the second lap meets the first, every one within the consistency bounds of the truth."""
    from loop_session import make_ring_session, run_session
    from oracle import oracle as orc

    kfs, order = make_ring_session(28, "VLP16", prefilter=lambda c: orc.voxelgrid(orc.distance_filter(c, 0.1, 35.0), 0.25, 1)[0])
    det = LoopDetector(registration=orc.Ndt(resolution=1.0, transformation_epsilon=0.01, maximum_iterations=64, num_threads=8))
    loops = run_session(det, kfs, order)
    assert len(loops) >= 2
    for lp in loops:
        assert lp.key1.accum_distance - lp.key2.accum_distance >= 15.0 or lp.key1.slam_uuid != lp.key2.slam_uuid


def test_store_key_names_the_cloud_content_not_just_the_id():
    """A keyframe whose cloud is replaced by one of EQUAL length, or another robot's keyframe with the same id, must not meet a stale
    resident copy in the GPU keyframe store (the reference re-reads candidate->cloud every call, loop_detector.cpp:128)."""
    a = _kf(7, 0, 0, 0.0)
    k0 = a.store_key()
    assert k0 == a.store_key() and 0 < k0 < (1 << 63)
    a.cloud = a.cloud.copy()
    assert a.store_key() == k0  # same content under the same name
    a.cloud = np.full((1, 4), 8, dtype=np.float32)  # equal length, other content
    assert a.store_key() != k0
    b = _kf(7, 0, 0, 0.0, uuid="b")
    assert b.store_key() != k0


def test_a_replaced_cloud_leaves_the_store_and_an_edited_one_raises():
    """the entry of a replaced cloud must not stay resident for ever, and an in-place edit of a hashed cloud must not meet the stale copy"""
    kfs = _chain(3)
    table = {(t, s): (None, True, 0.2) for t in (50, 51, 52) for s in range(1, 4)}
    m = ScriptedMatcher(table)
    det = LoopDetector({"enable_loop_closure_consistency_check": False}, matcher=m)
    det.matching(kfs[1:], _kf(50, 1.0, 0.0, 400.0))
    old_keys = set(m.store)
    assert len(old_keys) == 2
    replaced = kfs[1]
    old = replaced.store_key()
    replaced.cloud = replaced.cloud.copy() + np.float32(0.5)  # a NEW cloud object with other content: another name
    replaced.cloud[:, 0] = kfs[1].id  # (the scripted table reads the keyframe id out of x)
    det.matching(kfs[1:], _kf(51, 1.0, 0.0, 500.0))
    assert old not in m.store and replaced.store_key() in m.store and len(m.store) == 2  # the old entry was dropped, not leaked
    with pytest.raises(ValueError):
        replaced.cloud[0, 1] = 3.0  # hashed clouds are read-only: an in-place edit raises instead of meeting the resident copy


def _random_session(rng, n_known=40, n_new=24):
    """Two robots on two parallel tracks; graph edges along each track; `n_new` further keyframes of robot a arrive in groups."""
    kfs, per_robot = [], {"a": [], "b": []}
    for i in range(n_known):
        uuid = "a" if i % 3 else "b"
        lst = per_robot[uuid]
        kf = _kf(i + 1, 1.5 * len(lst), 0.0 if uuid == "a" else 2.0, 4.0 * len(lst), uuid=uuid, first=(len(lst) == 0))
        if rng.random() < 0.05:
            kf.static_keyframe = True
        if lst:
            prev = lst[-1]
            rel = np.linalg.inv(kf.estimate) @ prev.estimate
            kf.prev_edge = Edge(kf, prev, rel)
            prev.next_edge = Edge(kf, prev, rel)
        lst.append(kf)
        kfs.append(kf)
    news = [_kf(1000 + j, 1.5 * (j % 20) + rng.normal(0, 0.3), 1.0, 400.0 + 3.0 * j, uuid="a") for j in range(n_new)]
    table = {}
    for new in news:
        ne = normalize_estimate(new.estimate)
        for kf in kfs:
            # mostly near the true relative pose (consistent with the graph edges), now and then off by metres; scores around the 1.25 threshold
            dx = 0.0 if rng.random() < 0.7 else rng.normal(0, 1.5)
            T = (np.linalg.inv(ne) @ kf.estimate @ synth.make_pose([dx, 0, 0], synth.rot_z(0.0 if rng.random() < 0.8 else rng.normal(0, 0.1)))).astype(np.float32)
            score = float(rng.choice([0.2, 0.4, 0.4, 0.9, 1.3, 2.0])) if rng.random() < 0.6 else float(rng.uniform(0.1, 2.0))
            table[(new.id, kf.id)] = (T, bool(rng.random() < 0.85), score)
    return kfs, news, table


@pytest.mark.parametrize("seed", range(12))
def test_detect_batched_returns_the_sequential_loop_list(seed):
    """detect_batched (superset batch + consistency batch + host replay) against detect() — the reference's loop with its LoopManager gates
    (loop_detector.cpp:15-38,77-90) — on random two-robot sessions whose new keyframes arrive several per call: a loop found for one new keyframe
    prunes the candidates of the following ones (same-robot 15 m, other-robot 5 m of accumulated distance), ties, non-converged and over-threshold
    candidates, first / static keyframes, failed and rescued consistency checks all occur.  Same loops, same poses, and never an alignment outside
    the superset; the reference's counters (candidates per keyframe) come out the same."""
    rng = np.random.default_rng(1000 + seed)
    kfs, news, table = _random_session(rng)
    prm = {"accum_distance_thresh_same_robot": 15.0, "accum_distance_thresh_other_robot": 5.0, "candidate_max_xy_distance": 6.0}
    seq = LoopDetector(prm, registration=ScriptedRegistration(table))
    bat = LoopDetector(prm, matcher=ScriptedMatcher(table))
    group = [1, 2, 5, 8][seed % 4]
    loops_s, loops_b, pruned = [], [], 0
    for g0 in range(0, len(news), group):
        batch = news[g0:g0 + group]
        ls, lb = seq.detect(kfs, batch), bat.detect_batched(kfs, batch)
        assert [(lp.key1.id, lp.key2.id) for lp in ls] == [(lp.key1.id, lp.key2.id) for lp in lb]
        for a, b in zip(ls, lb):
            np.testing.assert_array_equal(a.relative_pose, b.relative_pose)
        loops_s += ls
        loops_b += lb
        pruned += bat.last_batched["superset_pairs"] - bat.last_batched["sequential_pairs"]
        assert bat.last_batched["superset_pairs"] >= bat.last_batched["sequential_pairs"]
    assert len(loops_s) >= 2
    assert seq.loop_candidates_sizes == bat.loop_candidates_sizes and len(bat.loop_detection_times) == len(bat.loop_candidates_sizes)
    assert seq.average_time_per_candidate_us() is not None and bat.average_time_per_candidate_us() is not None
    if group > 1:
        assert pruned > 0, "the session never exercised the LoopManager gates inside a call"
    # every alignment the sequential loop ran is in the batches (the batches may hold more: gated candidates, both consistency neighbours)
    assert set(seq.registration.calls) <= set(bat.matcher.calls)
    for name in ("a", "b"):
        la, lb_ = seq.loop_manager.get_loop("a", name), bat.loop_manager.get_loop("a", name)
        assert (la is None) == (lb_ is None) and (la is None or (la.key1.id, la.key2.id) == (lb_.key1.id, lb_.key2.id))


def test_detect_batched_edge_cases():
    kfs, news, table = _random_session(np.random.default_rng(5), n_known=12, n_new=3)
    bat = LoopDetector(matcher=ScriptedMatcher(table))
    assert bat.detect_batched(kfs, []) == [] and bat.detect_batched([], news) == []
    with pytest.raises(ValueError):
        LoopDetector(registration=ScriptedRegistration(table)).detect_batched(kfs, news)
