"""CPU, world_size 2 over gloo: the multi-GPU sharding / record gather / sequential best-candidate replay of
mrg_slam_amd/loop_closure.py gives the same answer as the single-process sequential loop
(/root/reference/src/mrg_slam/loop_detector.cpp:126-145)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _fake_records(n, seed=0):
    from mrg_slam_amd.registration import RESULT_DTYPE

    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=RESULT_DTYPE)
    r["T"] = rng.normal(size=(n, 16)).astype(np.float32)
    r["H"] = rng.normal(size=(n, 36))
    r["fitness"] = np.round(rng.uniform(0.1, 2.0, n), 1)  # coarse values: equal scores do occur
    r["converged"] = rng.uniform(size=n) > 0.25
    r["iterations"] = rng.integers(1, 30, n)
    r["evaluations"] = r["iterations"] * 3
    r["pair_id"] = np.arange(n)
    return r


class _FakeMatcher:
    """Stands in for BatchMatcher on the CPU: returns the precomputed record of every pair it was given."""

    def __init__(self, table):
        self.table, self.ids = table, []

    def add_target(self, cloud):
        return 0

    def add_pair(self, t, cloud, guess):
        self.ids.append(int(cloud[0, 0]))  # the fake "cloud" carries its candidate id

    def align(self, fitness_max_range):
        return self.table[self.ids].copy()


def _worker(rank, world, port, n, q):
    import torch.distributed as dist

    from mrg_slam_amd import loop_closure as lc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = _fake_records(n, seed=7)
    clouds = [np.full((1, 4), i, dtype=np.float32) for i in range(n)]
    rec, best, score = lc.match_candidates(lambda: _FakeMatcher(table), np.zeros((1, 4), np.float32), clouds, [np.eye(4)] * n)
    q.put((rank, rec.tobytes(), best, score, lc.shard_indices(n, world, rank).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [1, 2, 7, 256])
def test_two_rank_gather_equals_sequential_loop(n):
    from mrg_slam_amd import loop_closure as lc
    from mrg_slam_amd.registration import RESULT_DTYPE

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table = _fake_records(n, seed=7)
    exp_best, exp_score = lc.select_best(table)
    shards = set()
    for rank, blob, best, score, mine in outs:
        rec = np.frombuffer(blob, dtype=RESULT_DTYPE)
        assert rec.tobytes() == table.tobytes()  # every rank holds every record, in candidate order
        assert (best, score) == (exp_best, exp_score)
        assert mine == list(range(min(mine), max(mine) + 1)) if mine else True  # a contiguous block of the candidate list
        shards |= set(mine)
    assert shards == set(range(n))


def test_select_best_replays_reference_semantics():
    from mrg_slam_amd import loop_closure as lc
    from mrg_slam_amd.registration import RESULT_DTYPE

    r = np.zeros(5, dtype=RESULT_DTYPE)
    r["fitness"] = [0.5, 0.3, 0.3, 0.1, 0.3]
    r["converged"] = [1, 1, 1, 0, 1]
    assert lc.select_best(r) == (4, 0.3)  # "score > best_score" skips: the LAST of equal scores wins; non-converged never match
    r["converged"] = 0
    assert lc.select_best(r) == (None, np.finfo(np.float64).max)
    assert lc.select_best(r[:0]) == (None, np.finfo(np.float64).max)


def test_select_best_groups_equals_the_sequential_rule_per_group():
    """The vectorised pass over many new keyframes gives what select_best gives group by group: ragged and empty groups, equal scores, nothing
    converged, DBL_MAX and +inf scores, and NaN scores (which the reference's comparison lets through: replayed sequentially)."""
    from mrg_slam_amd import loop_closure as lc
    from mrg_slam_amd.registration import RESULT_DTYPE

    rng = np.random.default_rng(8)
    for trial in range(40):
        n = 60
        r = np.zeros(n, dtype=RESULT_DTYPE)
        r["fitness"] = rng.choice([0.1, 0.2, 0.2, 0.5, np.finfo(np.float64).max, np.inf], n)
        r["converged"] = rng.random(n) < 0.7
        if trial % 4 == 3:
            r["fitness"][rng.integers(0, n, 3)] = np.nan
        perm = rng.permutation(n)
        cuts = np.sort(rng.choice(np.arange(n + 1), 9, replace=True))
        groups = [list(perm[a:b]) for a, b in zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [n]]))]
        got = lc.select_best_groups(r, groups)
        exp = [lc.select_best(r[np.asarray(g, dtype=np.int64)]) for g in groups]
        assert len(got) == len(exp)
        for (gi, gs), (ei, es) in zip(got, exp):
            assert gi == ei and (gs == es or (np.isnan(gs) and np.isnan(es)))
    # single process (no process group): gather is the identity
    t = _fake_records(9)
    np.testing.assert_array_equal(lc.gather_records(t[::-1].copy(), 9), t)


def test_candidate_keys_reach_the_matcher_single_process():
    """match_candidates passes keyframe ids through to add_pair(key=...) (the keyframe store of the rank that owns the pair)."""
    from mrg_slam_amd import loop_closure as lc

    table = _fake_records(5, seed=3)
    seen = []

    class _KeyedMatcher(_FakeMatcher):
        def add_pair(self, t, cloud, guess, key=0):
            seen.append(key)
            super().add_pair(t, cloud, guess)

    clouds = [np.full((1, 4), i, dtype=np.float32) for i in range(5)]
    rec, best, score = lc.match_candidates(lambda: _KeyedMatcher(table), np.zeros((1, 4), np.float32), clouds, [np.eye(4)] * 5, candidate_keys=[11, 12, 13, 14, 15])
    assert seen == [11, 12, 13, 14, 15]
    rec2, best2, score2 = lc.match_candidates(lambda: _FakeMatcher(table), np.zeros((1, 4), np.float32), clouds, [np.eye(4)] * 5)
    assert rec.tobytes() == rec2.tobytes() and best == best2 and score == score2


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` without a launcher starts N ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) before touching
    the GPU, forwards rank 0's JSON line and fails when a rank fails (the flag used to be parsed and ignored)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SPAWN_TEST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["rank"] == 0 and line["master"] == "127.0.0.1" and int(line["port"]) > 0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, BENCH_SPAWN_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    # a rank that dies while another one waits (at the rendezvous, say): the launcher ends the waiting one instead of sitting out its timeout
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, BENCH_SPAWN_TEST_FAIL_RANK="1", BENCH_SPAWN_TEST_HANG_RANK="0"),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and time.time() - t0 < 60
    # under a launcher (WORLD_SIZE set) it does not spawn again
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 2


def test_loop_workload_is_config3_shaped_and_rank_independent():
    """bench.py --mode shard: 256 (new keyframe, candidate) pairs within 15 m on a 64-keyframe ring, seed 4242;
    the shards of G = 1, 2, 8 partition the same pair list and every target a rank builds is one its pairs use."""
    import importlib
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from mrg_slam_amd import loop_closure as lc
    from mrg_slam_amd import synth

    real = synth.synth_lidar_many
    synth.synth_lidar_many = lambda scene, poses, model, seeds, **kw: [np.zeros((1, 4), np.float32) for _ in poses]  # the scans themselves are not needed here
    try:
        scans, pairs = bench.make_loop_workload()
        scans2, pairs2 = bench.make_loop_workload()
    finally:
        synth.synth_lidar_many = real
    assert len(scans) == 64 and len(pairs) == 256
    assert all(a != b for a, b, _, _ in pairs) and [p[0] for p in pairs] == sorted(p[0] for p in pairs)
    for (a, b, g, rel), (a2, b2, g2, rel2) in zip(pairs, pairs2):
        assert (a, b) == (a2, b2) and np.array_equal(g, g2)
        assert np.linalg.norm(rel[:2, 3]) <= 15.0 + 1e-9 and np.linalg.norm(g[:3, 3] - rel[:3, 3]) < 3.0
    for world in (1, 2, 8):
        seen = np.concatenate([lc.shard_indices(256, world, r) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(256))
        assert max(len(lc.shard_indices(256, world, r)) for r in range(world)) == 256 // world
        # contiguous blocks of the keyframe-ordered list: a rank builds about 64 / world (+1) target grids, not all of them
        assert max(len({pairs[i][0] for i in lc.shard_indices(256, world, r)}) for r in range(world)) <= 64 // world + 2
    for n, world in ((7, 3), (5, 8), (0, 2)):
        parts = [lc.shard_indices(n, world, r) for r in range(world)]
        assert np.concatenate(parts).tolist() == list(range(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1
        assert sorted(np.concatenate([lc.shard_indices(n, world, r, "round_robin") for r in range(world)]).tolist()) == list(range(n))
