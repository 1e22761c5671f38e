"""GPU, several PROCESSES on one card: BASELINE config[4] (two robots, one process each, concurrent odometry streams on VLP-64 scans and
an inter-robot 64-candidate loop-closure batch, everything held against the CPU oracle run sequentially) and the two-rank record gather with
the real BatchMatcher.  The children are started as ordinary child processes (never an exec of this process)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_all(cmds, timeout):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    procs = [subprocess.Popen(c, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c in cmds]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def test_config4_two_robot_processes_at_stated_size(tmp_path):
    """BASELINE config[4]: 2 robots = 2 processes on the GPU at once (kitti_multirobot_processor.py:164-172), VLP-64 scans through the
    prefilter chain, 12 frames each with keyframe switches (scan_matching_odometry_component.cpp:326-339), then 64 inter-robot candidates
    per robot with getFitnessScore(inf).  Each process compares itself with the oracle's sequential run of the same loops."""
    frames = 12
    outs = _run_all([[sys.executable, os.path.join(ROOT, "tests", "workers", "robot_worker.py"), str(r), str(frames), str(tmp_path / f"robot{r}.json")] for r in (0, 1)], timeout=900)
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, e[-3000:])
        d = json.load(open(tmp_path / f"robot{r}.json"))
        assert d["points_per_filtered_scan"] > 20000  # VLP-64 scans after the 0.1 m voxel prefilter
        assert d["keyframes_gpu"] == d["keyframes_cpu"] and d["keyframes_gpu"] >= 4  # 1 m per scan against keyframe_delta_translation 1.0: several switches
        assert d["odometry_max_dt_m"] <= 1e-4 and d["odometry_max_dr_rad"] <= 1e-4, d
        assert d["odometry_same_iterations"], d
        assert d["batch_max_dt_m"] <= 1e-4 and d["batch_max_dr_rad"] <= 1e-4 and d["batch_mismatches"] == 0, d
        assert d["batch_max_rel_fitness_diff"] <= 1e-6, d
        assert d["batch_best_gpu"] == d["batch_best_cpu"], d
        assert d["final_odom_error_vs_truth_m"] < 0.5, d


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_gloo_ranks_with_the_real_matcher_equal_one_rank(tmp_path):
    """Two ranks share the card (gloo: RCCL refuses duplicate devices), each aligns its block of the candidate list with the real BatchMatcher,
    the records are all-gathered and the best candidate replayed: bit for bit the records and the answer of ONE rank matching all candidates."""
    n = 23
    worker = os.path.join(ROOT, "tests", "workers", "gloo_matcher_worker.py")
    one = _run_all([[sys.executable, worker, "0", "1", "0", str(n), str(tmp_path / "one.npy")]], timeout=600)
    assert one[0][0] == 0, one[0][2][-3000:]
    port = _free_port()
    two = _run_all([[sys.executable, worker, str(r), "2", str(port), str(n), str(tmp_path / "two.npy")] for r in (0, 1)], timeout=600)
    for rc, o, e in two:
        assert rc == 0, e[-3000:]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert len(a) == len(b) == n
    for f in ("T", "fitness", "converged", "iterations", "evaluations", "pair_id"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    np.testing.assert_allclose(a["H"], b["H"], rtol=1e-12)  # f64 sums carry the order of a round's items (1e-15 relative)
    assert one[0][1].strip().splitlines()[-1].split() == two[0][1].strip().splitlines()[-1].split()  # same best candidate and score (gloo chats on stdout before it)


def test_four_gloo_ranks_uneven_shards_equal_one_rank(tmp_path):
    """23 candidates over FOUR ranks sharing the card (blocks of 6 / 6 / 6 / 5): the all-gathered records and the replayed best candidate are
    those of one rank matching all 23 (padding records of the short shard never reach the replay)."""
    n = 23
    worker = os.path.join(ROOT, "tests", "workers", "gloo_matcher_worker.py")
    one = _run_all([[sys.executable, worker, "0", "1", "0", str(n), str(tmp_path / "one.npy")]], timeout=600)
    assert one[0][0] == 0, one[0][2][-3000:]
    port = _free_port()
    four = _run_all([[sys.executable, worker, str(r), "4", str(port), str(n), str(tmp_path / "four.npy")] for r in range(4)], timeout=600)
    for rc, o, e in four:
        assert rc == 0, e[-3000:]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "four.npy")
    assert len(a) == len(b) == n
    for f in ("T", "fitness", "converged", "iterations", "evaluations", "pair_id"):
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    np.testing.assert_allclose(a["H"], b["H"], rtol=1e-12)
    assert one[0][1].strip().splitlines()[-1].split() == four[0][1].strip().splitlines()[-1].split()


def test_rccl_process_group_of_one_rank_gathers_the_same_records(tmp_path):
    """The `backend="nccl"` (= RCCL) branch of loop_closure.gather_records — process-group init on the device, all_gather_into_tensor on DEVICE tensors,
    teardown — on the one-GPU box: a ONE-rank RCCL job through match_candidates returns the records of the job without a process group (that
    code had never run on hardware; reference for the independence being sharded: loop_detector.cpp:126-145)."""
    n = 11
    worker = os.path.join(ROOT, "tests", "workers", "gloo_matcher_worker.py")
    plain = _run_all([[sys.executable, worker, "0", "1", "0", str(n), str(tmp_path / "plain.npy")]], timeout=600)
    assert plain[0][0] == 0, plain[0][2][-3000:]
    rccl = _run_all([[sys.executable, worker, "0", "1", str(_free_port()), str(n), str(tmp_path / "rccl.npy"), "nccl"]], timeout=600)
    assert rccl[0][0] == 0, rccl[0][2][-3000:]
    assert "backend nccl world 1" in rccl[0][1]
    a, b = np.load(tmp_path / "plain.npy"), np.load(tmp_path / "rccl.npy")
    assert len(a) == len(b) == n and a.tobytes() == b.tobytes()
    assert plain[0][1].strip().splitlines()[-1].split() == rccl[0][1].strip().splitlines()[-1].split()


def test_bench_under_torchrun_one_rccl_rank(tmp_path):
    """The driver's SCALE launch line with N = 1 and the REAL backend: `python -m torch.distributed.run --nproc-per-node 1 … bench.py --gpus 1` — the launcher
    starts before anything touches the GPU; bench.py then initialises the RCCL process group, every collected batch's records go through
    all_gather_into_tensor on device tensors, the elapsed time through the MAX all-reduce, and rank 0 alone prints the compact line (< 4 KB).  Both
    modes: weak (config[1] shape, small batch) and shard (config[3], one step)."""
    env = dict(os.environ, BENCH_CACHE=str(tmp_path / "cache"), OMP_NUM_THREADS="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DIST_BACKEND"):
        env.pop(k, None)
    for extra, scaling in ((["--steps", "3", "--warmup", "1", "--batch", "16", "--shard-steps", "0"], "weak"), (["--mode", "shard", "--steps", "1", "--warmup", "1"], "strong")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu", "--no-extras", "--no-latency"] + extra
        run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert run.returncode == 0, run.stderr[-3000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) < 4096, lines
        line = json.loads(lines[0])
        assert line["n_gpus"] == 1 and line["scaling"] == scaling and line["value"] > 0 and line["config"]["record_gather"] == "nccl", line
        if scaling == "strong":
            assert line["config3"]["records_sha256_16"] and line["config3"]["pairs"] == 256


def test_bench_shard_mode_two_gloo_ranks_print_the_one_rank_digest(tmp_path):
    """`bench.py --mode shard` (BASELINE config[3], 256 pairs) as the driver's launcher would run it on two GPUs — here two gloo ranks on the one
    card (BENCH_DIST_BACKEND=gloo), started by bench.py's own launcher as child processes: the printed line says n_gpus 2, carries both ranks'
    phase times and the SAME record digest as the one-rank line."""
    env = dict(os.environ, BENCH_CACHE=str(tmp_path / "cache"), OMP_NUM_THREADS="8")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "shard", "--no-cpu", "--no-extras", "--steps", "1", "--warmup", "1", "--full-line"]
    one = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run(base + ["--gpus", "2"], env=dict(env, BENCH_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == "strong"
    s1, s2 = l1["config3_shard"], l2["config3_shard"]
    assert s1["pairs_total"] == s2["pairs_total"] == 256 and s2["pairs_per_gpu"] == 128
    assert s1["inputs_sha256_16"] == s2["inputs_sha256_16"] and s1["raw_inputs_as_in_the_build_container"] is True
    assert s1["records_sha256_16"] == s2["records_sha256_16"]
    assert [p["rank"] for p in s2["per_rank_phases_ms"]] == [0, 1] and all(p["alignment_rounds"] > 0 for p in s2["per_rank_phases_ms"])
    # the same 256 pairs through mrgfe_node_* (csrc/node.cpp): ONE process, three members sharing the card (blocks of 86 / 85 / 85), records gathered
    # behind the C ABI, best candidates from mrgfe_node_select_best — the one-rank digest again
    three = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "shard", "--inproc", "--gpus", "3", "--steps", "2", "--warmup", "1", "--full-line"], env=env,
                           capture_output=True, text=True, timeout=900)
    assert three.returncode == 0, three.stderr[-3000:]
    l3 = json.loads([ln for ln in three.stdout.splitlines() if ln.startswith("{")][-1])
    assert l3["n_gpus"] == 3 and l3["config"]["blocks"] == [[0, 86], [86, 85], [171, 85]] and l3["config"]["record_gather"] == "host"
    assert l3["inputs_sha256_16"] == s1["inputs_sha256_16"] and l3["records_sha256_16"] == s1["records_sha256_16"]


def test_bench_default_mode_under_torchrun_two_gloo_ranks(tmp_path):
    """The driver's SCALE launch line — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py --gpus N --steps K --warmup W`, the
    default (weak) mode with two batches in flight per rank and the record all-gather of every collected batch — with N = 2 gloo ranks on the one card
    (BENCH_DIST_BACKEND=gloo) and a small batch: ONE JSON line from rank 0, n_gpus 2, twice the pairs of a rank per step in `value`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_CACHE=str(tmp_path / "cache"), OMP_NUM_THREADS="8", BENCH_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "16", "--no-cpu", "--no-extras", "--shard-steps", "0"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert len(lines[0]) < 4096 and line["config"]["steps_in_flight"] == 2 and line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None
    assert abs(line["value"] - 2 * 16 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-4 * line["value"]  # (the line carries five significant digits)
    assert line["parity_vs_oracle"] is None or line["parity_vs_oracle"].get("pairs_over_bar", 0) == 0
