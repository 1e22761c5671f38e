"""Batched, multi-GPU loop-closure candidate matching: the host-side mirror of the candidate loop of
``LoopDetector::matching`` (/root/reference/src/mrg_slam/loop_detector.cpp:97-180).

The reference aligns every candidate keyframe against the new keyframe sequentially on one registration object
(:126-145) and keeps the candidate with the lowest fitness score among the converged ones.  The alignments are
independent (they share only the read-only target), so here

* the pairs are sharded over the ranks (one process per GPU) in contiguous blocks of the candidate list, sizes differing by at
  most one.  The list is ordered by new keyframe (the reference matches keyframe after keyframe,
  loop_detector.cpp:21-31), so a rank's block touches few distinct targets and builds only those voxel grids — with round-robin
  (pair i -> rank i mod G) every rank built nearly every target of a many-keyframe batch (64 / 64 / 53 / 32 of 64 at G = 1 / 2 / 4 / 8);
* each rank advances its shard on its GPU with the batched engine (``BatchMatcher``), no data-path collective,
* the ranks all-gather the fixed-size 384-byte result records (pose, 6x6 Hessian, fitness, flags) — RCCL over xGMI
  on GPUs (``backend="nccl"``), gloo in the CPU tests,
* every rank replays the reference's sequential selection on the gathered records, so the answer is identical to the
  sequential loop whatever the number of GPUs.

Only this module touches ``torch.distributed``; the records are plain numpy structured arrays (``RESULT_DTYPE``).
"""
from __future__ import annotations

import numpy as np

from .registration import RESULT_DTYPE


def shard_indices(n_pairs: int, world_size: int, rank: int, policy: str = "block") -> np.ndarray:
    """Pairs owned by ``rank``.  "block" (default): the rank-th of world_size contiguous blocks whose sizes differ by at most one
    (the first n_pairs mod world_size blocks are the longer ones); "round_robin": i with i mod world_size == rank."""
    if policy == "round_robin":
        return np.arange(rank, n_pairs, world_size, dtype=np.int64)
    base, extra = divmod(n_pairs, world_size)
    lo = rank * base + min(rank, extra)
    return np.arange(lo, lo + base + (1 if rank < extra else 0), dtype=np.int64)


def gather_records(local: np.ndarray, n_pairs: int, group=None) -> np.ndarray:
    """All-gather the per-rank result records and return them ordered by global pair id (``pair_id`` field).

    ``local`` holds this rank's records with ``pair_id`` already set to the GLOBAL pair index.  Shards are padded to the
    largest shard so one fixed-size collective suffices (ceil(n_pairs / G) * 384 bytes per rank)."""
    import torch
    import torch.distributed as dist

    if local.dtype != RESULT_DTYPE:
        raise TypeError("records must use RESULT_DTYPE")
    if not dist.is_available() or not dist.is_initialized():
        out = np.zeros(n_pairs, dtype=RESULT_DTYPE)
        out[local["pair_id"]] = local
        return out
    world = dist.get_world_size(group)
    per = -(-n_pairs // world)
    pad = np.zeros(per, dtype=RESULT_DTYPE)
    pad["pair_id"] = -1
    pad[: len(local)] = local
    use_cuda = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    mine = torch.from_numpy(pad.view(np.uint8).reshape(per, RESULT_DTYPE.itemsize).copy()).to(dev)
    everyone = torch.empty((world * per, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    rec = np.frombuffer(everyone.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
    rec = rec[rec["pair_id"] >= 0]
    out = np.zeros(n_pairs, dtype=RESULT_DTYPE)
    out[rec["pair_id"]] = rec
    return out


def select_best(records: np.ndarray):
    """Replay of loop_detector.cpp:126-145 on gathered records, in candidate order:

        if (!hasConverged() || score > best_score) continue;   best_score = score; best_matched = candidate;

    so among equal scores the LAST candidate wins and non-converged candidates never match.
    Returns (best_index or None, best_score)."""
    best_score = np.finfo(np.float64).max
    best = None
    for i in range(len(records)):
        score = float(records[i]["fitness"])
        if not records[i]["converged"] or score > best_score:
            continue
        best_score = score
        best = i
    return best, best_score


def select_best_groups(records: np.ndarray, groups):
    """:func:`select_best` for many new keyframes at once: ``groups[g]`` = indices into ``records`` of keyframe g's candidates in candidate order.
    Returns a list of (best position within the group or None, best score), the same values the sequential rule gives group by group — one
    vectorised pass when no score is NaN (the rule's `score > best_score` lets a NaN through and then accepts everything after it: that case
    is replayed group by group)."""
    n_g = len(groups)
    if n_g == 0:
        return []
    width = max(len(g) for g in groups)
    if width == 0:
        return [(None, np.finfo(np.float64).max)] * n_g
    idx = np.zeros((n_g, width), dtype=np.int64)
    valid = np.zeros((n_g, width), dtype=bool)
    for k, g in enumerate(groups):
        idx[k, : len(g)] = g
        valid[k, : len(g)] = True
    return _select_best_table(records, idx, valid, groups)


def _select_best_table(records, idx, valid, groups):
    fit = records["fitness"][idx].astype(np.float64)
    ok = valid & (records["converged"][idx] != 0)
    if np.isnan(fit[ok]).any():
        return [select_best(records[np.asarray(g, dtype=np.int64)]) for g in groups]
    big = np.finfo(np.float64).max
    # candidates the rule can accept at all: converged and score <= the initial best_score (DBL_MAX): +inf never matches
    ok &= fit <= big
    score = np.where(ok, fit, np.inf)
    best = score.min(axis=1)
    hit = ok & (score == best[:, None])
    last = idx.shape[1] - 1 - np.argmax(hit[:, ::-1], axis=1)  # among equal scores the LAST candidate wins
    some = hit.any(axis=1)
    return [(int(last[k]), float(best[k])) if some[k] else (None, big) for k in range(len(groups))]


def match_candidates(matcher_factory, target_cloud, candidate_clouds, guesses, fitness_max_range=float("inf"), group=None, candidate_keys=None):
    """Distributed candidate matching for one new keyframe.

    ``matcher_factory()`` returns a ``BatchMatcher`` bound to this rank's GPU (a fresh one, or — to profit from the keyframe
    store — the same cleared one every call); ``target_cloud`` is the new keyframe's cloud
    (registration_->setInputTarget(new_keyframe->cloud), :104), ``candidate_clouds[i]`` / ``guesses[i]`` the candidates and
    their initial guesses (:127-133).  ``candidate_keys[i]`` (optional, non-zero keyframe ids) keep the candidates' clouds
    and GICP covariances resident on the rank that matched them.  Returns (records ordered by candidate, best index, best score)."""
    import torch.distributed as dist

    n = len(candidate_clouds)
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    mine = shard_indices(n, world, rank)
    local = np.zeros(len(mine), dtype=RESULT_DTYPE)
    if len(mine):
        bm = matcher_factory()
        t = bm.add_target(target_cloud)
        for i in mine:
            if candidate_keys is None:
                bm.add_pair(t, candidate_clouds[i], guesses[i])
            else:
                bm.add_pair(t, candidate_clouds[i], guesses[i], key=int(candidate_keys[i]))
        local = bm.align(fitness_max_range)
        local["pair_id"] = mine.astype(np.int32)
    records = gather_records(local, n, group)
    best, score = select_best(records)
    return records, best, score
