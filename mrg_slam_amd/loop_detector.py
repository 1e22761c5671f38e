"""Host mirror of ``mrg_slam::LoopDetector`` (/root/reference/src/mrg_slam/loop_detector.cpp, include/mrg_slam/loop_detector.hpp): the
caller of the hot path in the back end — ``find_candidates`` (:41-95), the candidate loop of ``matching`` (:97-180) with the planar
guess (:129-133), the best-score rule (:137-144), the ``fitness_score_thresh`` gate (:156-160), the consistency check against the best
match's previous / next keyframe (:190-303) and the ``LoopManager`` bookkeeping.

Two ways to run ``matching`` on the same decisions:

* ``registration=`` anything with the ``pcl::Registration`` call surface (a HIP registration or the CPU oracle's classes): the
  reference's loop as it stands, one ``align`` at a time;
* ``matcher=`` a ``BatchMatcher``: all candidates of a new keyframe advance together on the GPU (``mrgfe_batch_*``, candidates named by
  keyframe id so that their clouds stay in the HBM keyframe store), then a second batch of at most two pairs — the best match's previous and
  next keyframe against the same target — supplies the consistency check; the next keyframe's result is used only where the reference
  would have computed it (the previous one failed).  The alignments are independent, so the Loop list is the reference's.

``detect_batched`` goes one level up: ``detect`` (:15-38) calls ``matching`` once per NEW keyframe, 1-30 candidates each — launch-latency-sized
batches.  The LoopManager gates of ``find_candidates`` (:77-90) only REMOVE candidates, so the candidates of all new keyframes with the gates
ignored are a superset of what the sequential loop will ever align: ONE batch aligns and scores them all (one target grid per new keyframe), a
second batch supplies the consistency alignments of every best match any gate state can produce, and the gates, the best-score rule, the
thresholds and ``add_loop`` are replayed keyframe after keyframe on the host from the records.  Same Loop list as ``detect``.

Poses are ``numpy`` 4 x 4 matrices: keyframe estimates in double (``Eigen::Isometry3d``), registration results in float
(``Eigen::Matrix4f``), products and inverses in the type the reference forms them in."""
from __future__ import annotations

import dataclasses
import hashlib
import time

import numpy as np

from . import loop_closure

DEFAULTS = {  # config/mrg_slam.yaml:169-179
    "candidate_max_xy_distance": 15.0,
    "accum_distance_thresh_same_robot": 15.0,
    "accum_distance_thresh_other_robot": 5.0,
    "fitness_score_max_range": float("inf"),
    "fitness_score_thresh": 1.25,
    "use_planar_registration_guess": False,
    "enable_loop_closure_consistency_check": True,
    "loop_closure_consistency_max_delta_trans": 0.3,
    "loop_closure_consistency_max_delta_angle": 0.0523599,
}


@dataclasses.dataclass(eq=False)
class Edge:
    """The slice of mrg_slam::Edge the loop detector reads: ``from_keyframe`` --relative_pose--> ``to_keyframe``."""
    from_keyframe: "KeyFrame"
    to_keyframe: "KeyFrame"
    relative_pose: np.ndarray  # 4 x 4 double


@dataclasses.dataclass(eq=False)
class KeyFrame:
    """The slice of mrg_slam::KeyFrame the loop detector reads (include/mrg_slam/keyframe.hpp)."""
    id: int                      # non-zero: names the cloud in the GPU keyframe store
    cloud: np.ndarray            # N x 4 float32
    estimate: np.ndarray         # node->estimate(): 4 x 4 double
    accum_distance: float
    slam_uuid: str = "robot"
    first_keyframe: bool = False
    static_keyframe: bool = False
    prev_edge: Edge | None = None  # this keyframe -> the one before it (prev_edge->to_keyframe, loop_detector.cpp:213)
    next_edge: Edge | None = None  # the one after it -> this keyframe (next_edge->from_keyframe, :256)
    connected: set = dataclasses.field(default_factory=set)  # ids of keyframes this one shares a graph edge with (KeyFrame::edge_exists)

    def edge_exists(self, other: "KeyFrame") -> bool:
        return other.id in self.connected

    def store_key(self) -> int:
        """Name of this keyframe's cloud in the GPU keyframe store: a non-zero 63-bit hash of (slam_uuid, id, cloud content).

        The reference re-reads ``candidate->cloud`` on every ``matching`` call (loop_detector.cpp:128), so a keyframe whose cloud was
        replaced — or two robots reusing an id — must not meet a stale resident copy: the content is part of the name.  The digest is
        computed once per cloud OBJECT (keyframe clouds are ``ConstPtr`` in the reference: replaced, never edited in place), and the array is
        made read-only when it is hashed, so that an in-place edit raises instead of silently meeting the resident copy of the old content.
        ``retired_store_key`` names the entry of the cloud this one replaced (once): the detector drops it from the GPU store."""
        tok = getattr(self, "_store_token", None)
        if tok is None or tok[0] is not self.cloud:
            if tok is not None:
                self._retired_key = tok[1]
            if isinstance(self.cloud, np.ndarray):
                self.cloud.setflags(write=False)
            c = np.ascontiguousarray(self.cloud, dtype=np.float32)
            h = hashlib.blake2b(digest_size=8)
            h.update(f"{self.slam_uuid}/{self.id}/{c.shape[0]}/".encode())
            h.update(c.tobytes())
            tok = (self.cloud, (int.from_bytes(h.digest(), "little") & ((1 << 63) - 1)) | 1)
            self._store_token = tok
        return tok[1]

    def retired_store_key(self):
        """The store key of the cloud that ``cloud`` replaced, once (None otherwise): its resident copy is garbage now."""
        k = getattr(self, "_retired_key", None)
        self._retired_key = None
        return k


@dataclasses.dataclass(eq=False)
class Loop:
    key1: KeyFrame             # new keyframe testing for loop closure
    key2: KeyFrame             # best matched candidate keyframe
    relative_pose: np.ndarray  # float 4 x 4, key1 -> key2


class LoopManager:
    """loop_detector.hpp:39-115: the most recent loop per (new keyframe's SLAM instance, candidate's SLAM instance)."""

    def __init__(self):
        self.loop_map: dict = {}

    def get_loop(self, new_uuid, cand_uuid):
        return self.loop_map.get(new_uuid, {}).get(cand_uuid)

    def add_loop(self, loop: Loop):
        self.loop_map.setdefault(loop.key1.slam_uuid, {})[loop.key2.slam_uuid] = loop


def _quat_from_matrix(R, dtype):
    """Eigen's matrix -> quaternion conversion (w, x, y, z) in ``dtype``."""
    R = np.asarray(R, dtype=dtype)
    one, half = dtype(1), dtype(0.5)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4, dtype=dtype)
    if t > 0:
        t = np.sqrt(t + one)
        q[0] = half * t
        t = half / t
        q[1], q[2], q[3] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + one)
        q[1 + i] = half * t
        t = half / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def _quat_to_matrix(q):
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]], dtype=q.dtype)


def normalize_estimate(estimate) -> np.ndarray:
    """LoopDetector::normalize_estimate (:183-188): linear() = Quaterniond(linear()).normalized().toRotationMatrix()."""
    out = np.array(estimate, dtype=np.float64)
    q = _quat_from_matrix(out[:3, :3], np.float64)
    out[:3, :3] = _quat_to_matrix(q / np.linalg.norm(q))
    return out


def angular_distance_to_identity(R) -> float:
    """Eigen::Quaternionf(R).angularDistance(Quaternionf::Identity()) = 2 atan2(|vec|, |w|), in float."""
    q = _quat_from_matrix(R, np.float32)
    return float(np.float32(2.0) * np.arctan2(np.linalg.norm(q[1:]).astype(np.float32), np.abs(q[0])))


class LoopDetector:
    def __init__(self, params: dict | None = None, registration=None, matcher=None):
        if (registration is None) == (matcher is None):
            raise ValueError("give either a registration object (the reference's sequential loop) or a BatchMatcher (all candidates at once)")
        self.p = dict(DEFAULTS)
        self.p.update(params or {})
        self.registration, self.matcher = registration, matcher
        self.loop_manager = LoopManager()
        self.alignments = 0  # registrations run so far (candidates + consistency checks)
        # the reference's only performance counters (loop_detector.cpp:22-34, written out as average_time_per_candidate_us by
        # apps/mrg_slam_component.cpp:1032-1037): candidates and microseconds per new keyframe that had candidates
        self.loop_candidates_sizes: list = []
        self.loop_detection_times: list = []

    # ---- detect (:14-38)
    def detect(self, keyframes, new_keyframes):
        detected = []
        for new_keyframe in new_keyframes:
            start = time.perf_counter()
            candidates = self.find_candidates(new_keyframe, keyframes)
            loop = self.matching(candidates, new_keyframe)
            if loop is not None:
                detected.append(loop)
            if candidates:  # :29-33
                self.loop_candidates_sizes.append(len(candidates))
                self.loop_detection_times.append(int(1e6 * (time.perf_counter() - start)))
        return detected

    def average_time_per_candidate_us(self) -> float | None:
        """apps/mrg_slam_component.cpp:1032-1037: total loop detection time over total candidates."""
        total = sum(self.loop_candidates_sizes)
        return float(sum(self.loop_detection_times)) / total if total else None

    # ---- detect, all new keyframes in one batch
    def detect_batched(self, keyframes, new_keyframes):
        """``detect`` with the alignments of ALL new keyframes in two batches (module docstring).  Needs ``matcher=``.  Returns the Loop list
        ``detect(keyframes, new_keyframes)`` returns, loop for loop (same keyframes, same relative poses)."""
        if self.matcher is None:
            raise ValueError("detect_batched needs a BatchMatcher (matcher=)")
        from .registration import result_matrix

        start = time.perf_counter()
        p, bm = self.p, self.matcher
        new_keyframes = list(new_keyframes)
        new_est = [normalize_estimate(k.estimate) for k in new_keyframes]
        supersets = [self.find_candidates(k, keyframes, ignore_loop_manager=True) for k in new_keyframes]
        n_super = sum(len(s) for s in supersets)
        if n_super == 0:
            return []

        def queue(pairs):  # pairs: (index of the new keyframe, source keyframe); one target grid per new keyframe that has pairs
            bm.clear()
            tid = {}
            for k, kf in pairs:
                if k not in tid:
                    tid[k] = bm.add_target(new_keyframes[k].cloud)  # registration_->setInputTarget(new_keyframe->cloud), :104
                key = kf.store_key()
                old = kf.retired_store_key()
                if old is not None and old != key:
                    bm.forget(old)
                have = bm.has_cloud(key) == len(kf.cloud)
                bm.add_pair(tid[k], None if have else kf.cloud, self._guess(new_est[k], kf), key=key)
            self.alignments += len(pairs)

        # ---- batch 1: every (new keyframe, superset candidate) pair, aligned and scored
        pairs1 = [(k, c) for k, cands in enumerate(supersets) for c in cands]
        queue(pairs1)
        rec1 = bm.align(p["fitness_score_max_range"])
        first = np.cumsum([0] + [len(s) for s in supersets])
        results = [[(result_matrix(r), bool(r["converged"]), float(r["fitness"])) for r in rec1[first[k]:first[k + 1]]] for k in range(len(new_keyframes))]

        def best_of(k, keep):  # the rule of :137-144 over the candidates `keep` leaves, in candidate order
            best_score, best_matched, rel_pose = np.finfo(np.float64).max, None, None
            for candidate, (T, converged, score), use in zip(supersets[k], results[k], keep):
                if not use or not converged or score > best_score:
                    continue
                best_score, best_matched, rel_pose = score, candidate, T
            return best_score, best_matched, rel_pose

        def needs_alignments(best_matched, best_score):  # _consistency_check reaches the registration
            return (best_matched is not None and not (best_matched.first_keyframe or best_matched.static_keyframe) and p["enable_loop_closure_consistency_check"]
                    and not best_score > p["fitness_score_thresh"])

        # ---- batch 2: the gates drop ALL candidates of a SLAM instance or none (their conditions name the candidate's instance only), so the candidate
        # lists the replay can meet are the supersets minus any set of instances: the best match of each is known now, and its previous / next keyframe
        # are aligned against the new keyframe for all of them at once
        pairs2, seen2 = [], set()
        for k, cands in enumerate(supersets):
            robots = sorted({c.slam_uuid for c in cands})
            for mask in range(1 << len(robots)):
                gated = {r for b, r in enumerate(robots) if mask >> b & 1}
                best_score, best_matched, _ = best_of(k, [c.slam_uuid not in gated for c in cands])
                if not needs_alignments(best_matched, best_score):
                    continue
                for kf in (best_matched.prev_edge.to_keyframe if best_matched.prev_edge is not None else None,
                           best_matched.next_edge.from_keyframe if best_matched.next_edge is not None else None):
                    if kf is not None and (k, id(kf)) not in seen2:
                        seen2.add((k, id(kf)))
                        pairs2.append((k, kf))
        T2 = {}
        if pairs2:
            queue(pairs2)
            rec2 = bm.align(-1.0)
            T2 = {(k, id(kf)): result_matrix(r) for (k, kf), r in zip(pairs2, rec2)}

        # ---- the reference's loop, keyframe after keyframe, on the records
        detected, kept_sizes = [], []
        for k, new_keyframe in enumerate(new_keyframes):
            keep = [not self._loop_manager_gate(new_keyframe, c) for c in supersets[k]]
            n_cand = sum(keep)
            if n_cand == 0:
                continue
            kept_sizes.append(n_cand)
            best_score, best_matched, rel_pose = best_of(k, keep)
            consistent = False
            if best_matched is not None and (best_matched.first_keyframe or best_matched.static_keyframe):
                consistent = True
            elif needs_alignments(best_matched, best_score):
                prev_kf = best_matched.prev_edge.to_keyframe if best_matched.prev_edge is not None else None
                next_kf = best_matched.next_edge.from_keyframe if best_matched.next_edge is not None else None
                consistent = bool((prev_kf is not None and self._consistent_prev(best_matched, rel_pose, T2[(k, id(prev_kf))]))
                                  or (next_kf is not None and self._consistent_next(best_matched, rel_pose, T2[(k, id(next_kf))])))
            if best_score > p["fitness_score_thresh"]:
                continue
            if p["enable_loop_closure_consistency_check"] and best_matched is not None and not best_matched.first_keyframe and not consistent:
                continue
            loop = Loop(new_keyframe, best_matched, rel_pose)
            self.loop_manager.add_loop(loop)
            detected.append(loop)
        # the reference's counters: the candidates the sequential loop would have aligned; the call's time shared out by candidate count
        elapsed_us = 1e6 * (time.perf_counter() - start)
        total = sum(kept_sizes)
        for n in kept_sizes:
            self.loop_candidates_sizes.append(n)
            self.loop_detection_times.append(int(elapsed_us * n / total))
        self.last_batched = {"superset_pairs": n_super, "sequential_pairs": total, "consistency_pairs": len(pairs2), "new_keyframes": len(new_keyframes)}
        return detected

    # ---- find_candidates (:41-95)
    def _loop_manager_gate(self, new_keyframe: KeyFrame, candidate: KeyFrame) -> bool:
        """:77-90, True = the candidate is skipped: a loop was closed too recently (in accumulated distance) between the two SLAM instances.
        Depends on the candidate through its ``slam_uuid`` only."""
        p = self.p
        last_loop = self.loop_manager.get_loop(new_keyframe.slam_uuid, candidate.slam_uuid)
        if last_loop and new_keyframe.slam_uuid == candidate.slam_uuid and new_keyframe.accum_distance - last_loop.key1.accum_distance < p["accum_distance_thresh_same_robot"]:
            return True
        if last_loop and new_keyframe.slam_uuid != candidate.slam_uuid and new_keyframe.accum_distance - last_loop.key1.accum_distance < p["accum_distance_thresh_other_robot"]:
            return True
        return False

    def find_candidates(self, new_keyframe: KeyFrame, keyframes, ignore_loop_manager: bool = False):
        p = self.p
        max_sq = p["candidate_max_xy_distance"] * p["candidate_max_xy_distance"]
        out = []
        for candidate in keyframes:
            if new_keyframe.edge_exists(candidate):  # there is already an edge
                continue
            if candidate.first_keyframe:  # first keyframes don't filter out points that hit other rovers
                continue
            d = candidate.estimate[:2, 3] - new_keyframe.estimate[:2, 3]
            if float(d @ d) > max_sq:
                continue
            if new_keyframe.slam_uuid == candidate.slam_uuid and new_keyframe.accum_distance - candidate.accum_distance < p["accum_distance_thresh_same_robot"]:
                continue
            if not ignore_loop_manager and self._loop_manager_gate(new_keyframe, candidate):
                continue
            out.append(candidate)
        return out

    def _guess(self, new_estimate, kf: KeyFrame) -> np.ndarray:
        g = (np.linalg.inv(new_estimate) @ normalize_estimate(kf.estimate)).astype(np.float32)  # :130-133
        if self.p["use_planar_registration_guess"]:
            g[2, 3] = 0.0
        return g

    # ---- the alignments: (final transformation, converged, fitness) per source against the new keyframe's cloud
    def _align_all(self, new_keyframe: KeyFrame, sources, guesses, want_fitness: bool):
        self.alignments += len(sources)
        max_range = self.p["fitness_score_max_range"]
        if self.registration is not None:
            out = []
            for kf, g in zip(sources, guesses):
                self.registration.setInputSource(kf.cloud)
                self.registration.align(g)
                score = self.registration.getFitnessScore(max_range) if want_fitness else None
                out.append((np.asarray(self.registration.getFinalTransformation(), dtype=np.float32), bool(self.registration.hasConverged()), score))
            return out
        from .registration import result_matrix

        bm = self.matcher
        bm.clear()
        t = bm.add_target(new_keyframe.cloud)
        for kf, g in zip(sources, guesses):
            key = kf.store_key()  # (slam_uuid, id, content): an equal-length replacement or another robot's same id is another entry
            old = kf.retired_store_key()
            if old is not None and old != key:
                bm.forget(old)  # the replaced cloud's resident copy (the batch was just cleared: nothing references it)
            have = bm.has_cloud(key) == len(kf.cloud)
            bm.add_pair(t, None if have else kf.cloud, g, key=key)
        res = bm.align(max_range if want_fitness else -1.0)
        return [(result_matrix(r), bool(r["converged"]), float(r["fitness"]) if want_fitness else None) for r in res]

    # ---- matching (:97-180)
    def matching(self, candidate_keyframes, new_keyframe: KeyFrame):
        if not candidate_keyframes:
            return None
        if self.registration is not None:
            self.registration.setInputTarget(new_keyframe.cloud)  # :104
        new_estimate = normalize_estimate(new_keyframe.estimate)
        guesses = [self._guess(new_estimate, c) for c in candidate_keyframes]
        results = self._align_all(new_keyframe, candidate_keyframes, guesses, True)
        best_score, best_matched, rel_pose = np.finfo(np.float64).max, None, None
        for candidate, (T, converged, score) in zip(candidate_keyframes, results):  # :137-144
            if not converged or score > best_score:
                continue
            best_score, best_matched, rel_pose = score, candidate, T
        consistent = self._consistency_check(new_keyframe, new_estimate, rel_pose, best_matched, best_score)
        if best_score > self.p["fitness_score_thresh"]:  # :156-160
            return None
        if self.p["enable_loop_closure_consistency_check"] and best_matched is not None and not best_matched.first_keyframe and not consistent:  # :162-166
            return None
        loop = Loop(new_keyframe, best_matched, rel_pose)
        self.loop_manager.add_loop(loop)
        return loop

    # ---- perform_loop_closure_consistency_check (:190-210) with :212-303
    def _consistency_check(self, new_keyframe, new_estimate, rel_pose_new_to_best, best_matched, best_score) -> bool:
        p = self.p
        if best_matched is not None and (best_matched.first_keyframe or best_matched.static_keyframe):
            return True
        if best_matched is None or not p["enable_loop_closure_consistency_check"] or best_score > p["fitness_score_thresh"]:
            return False
        prev_kf = best_matched.prev_edge.to_keyframe if best_matched.prev_edge is not None else None
        next_kf = best_matched.next_edge.from_keyframe if best_matched.next_edge is not None else None
        if self.registration is not None:  # the reference's order: the next keyframe only if the previous one failed
            if prev_kf is not None and self._consistent_prev(best_matched, rel_pose_new_to_best, self._align_all(new_keyframe, [prev_kf], [self._guess(new_estimate, prev_kf)], False)[0][0]):
                return True
            return next_kf is not None and self._consistent_next(best_matched, rel_pose_new_to_best, self._align_all(new_keyframe, [next_kf], [self._guess(new_estimate, next_kf)], False)[0][0])
        # batched: both neighbours in one launch; the decisions are evaluated in the reference's order
        srcs = [kf for kf in (prev_kf, next_kf) if kf is not None]
        if not srcs:
            return False
        res = self._align_all(new_keyframe, srcs, [self._guess(new_estimate, kf) for kf in srcs], False)
        T = {id(kf): r[0] for kf, r in zip(srcs, res)}
        if prev_kf is not None and self._consistent_prev(best_matched, rel_pose_new_to_best, T[id(prev_kf)]):
            return True
        return next_kf is not None and self._consistent_next(best_matched, rel_pose_new_to_best, T[id(next_kf)])

    def _within(self, M) -> bool:
        delta_trans = float(np.linalg.norm(M[:3, 3].astype(np.float32)))
        delta_angle = angular_distance_to_identity(M[:3, :3])
        return not (delta_trans > self.p["loop_closure_consistency_max_delta_trans"] or delta_angle > self.p["loop_closure_consistency_max_delta_angle"])

    def _consistent_prev(self, best_matched, rel_pose_new_to_best, rel_pose_new_to_prev) -> bool:
        cand_to_prev = best_matched.prev_edge.relative_pose.astype(np.float32)  # :218
        M = (np.linalg.inv(rel_pose_new_to_prev.astype(np.float32)).astype(np.float32) @ rel_pose_new_to_best.astype(np.float32) @ cand_to_prev).astype(np.float32)  # :233
        return self._within(M)

    def _consistent_next(self, best_matched, rel_pose_new_to_best, rel_pose_new_to_next) -> bool:
        next_to_cand = best_matched.next_edge.relative_pose.astype(np.float32)  # :261
        M = (np.linalg.inv(rel_pose_new_to_best.astype(np.float32)).astype(np.float32) @ rel_pose_new_to_next.astype(np.float32) @ next_to_cand).astype(np.float32)  # :276
        return self._within(M)


__all__ = ["DEFAULTS", "Edge", "KeyFrame", "Loop", "LoopManager", "LoopDetector", "normalize_estimate", "angular_distance_to_identity", "loop_closure"]
