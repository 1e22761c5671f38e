#!/usr/bin/env python3
"""bench.py — scan-pair NDT alignments per second on MI355X (BASELINE.json metric), one rank per GPU.

Step      = one pass of the hot path over one batch of B (default 256: the candidate-batch size of BASELINE config[3])
            DISTINCT synthetic VLP-64 scan pairs already resident in HBM (2 x 257 scans ~ 1 GB per rank: past the 256 MiB
            Infinity Cache, so the source stream of every derivative launch really comes from HBM):
            for every pair  setInputTarget (voxel covariance grid build)  +  setInputSource  +  align(guess)
            (reference call sites: apps/scan_matching_odometry_component.cpp:203,208,265-266; loop_detector.cpp:104,127,134),
            advanced together by the batched engine (one derivative launch per round for all pairs still running).
Workload  = BASELINE config[1] shape: ~120k points per scan, NDT resolution 1.0 m, DIRECT7, reg_transformation_epsilon 0.1,
            reg_maximum_iterations 64 (config/mrg_slam.yaml:100-109); initial guesses as SURVEY.md §8(d) specifies them: warm
            (perturbed truth, seed 777+k) for three pairs in four, cold (identity) for every fourth (--cold-every).
            No KITTI data exists here: scans are ray-cast by mrg_slam_amd/synth.py; the 0.1 m voxel
            prefilter saturates this synthetic street at ~35k points, so the ~120k-point clouds the metric is quoted
            on are the distance-filtered (0.1..35 m) scans (--prefilter full selects the whole chain instead).
N > 1     = `python bench.py --gpus N` starts the N ranks itself (child processes, before anything touches the GPU) unless a
            launcher (torchrun) already did (WORLD_SIZE set).
            --mode weak (default): every rank aligns its own B pairs (independent units, no data-path collective), then the
            ranks all-gather the 384-byte result records over RCCL (the pose/Hessian gather of the north star).
            --mode shard: BASELINE config[3] as stated — 256 loop-closure candidate pairs in total (64 keyframes on a 40 m
            ring, seed 4242), contiguous blocks of the keyframe-ordered pair list per rank, every rank builds only the target grids its pairs need, fitness score
            with max_range = inf, RCCL all-gather of the records, replay of the reference's best-candidate rule
            (loop_detector.cpp:126-145) per new keyframe.  Strong scaling.  The default run also measures a few steps of it
            and reports them under "config3_shard" (never as `value`).
Timed     = K steps with TWO batches in flight (--in-flight 2, round 5: mrgfe_batch_align_async / _wait on two contexts; a step submits its batch
region      and collects the one submitted two steps earlier; everything still in flight is collected before the closing synchronisation).  The
            same K steps one at a time are measured beside it (`value_one_step_at_a_time`; --in-flight 1 makes that the timed region).
Output    = ONE JSON line (rank 0) with `roofline` (derivative kernel: algorithmic bytes / HIP-event time on the launch
            stream; under pipelining two launches share the chip: `aggregate_*` and `one_step_at_a_time` beside the per-launch figure) and
            `cpu_baseline` (the CPU oracle, kind "port", timed on a bounded sample of the same pairs).  Side measurements, never `value`:
            `value_host_pointers` (both clouds of every pair over PCIe inside the step: pageable, page-locked, page-locked + two batches in
            flight), `gpu_split_ms_per_step`, `pipeline_shape`, `config2_gicp`, `pcl_ndt`, `config3_shard` (+ `two_batches_in_flight`), `soak_over_bar`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "scan-pair alignments/sec (NDT, ~120k pts, 1.0 m voxel)"


def rot_angle(Ra, Rb):
    from mrg_slam_amd import synth

    return synth.rotation_angle(Ra, Rb)


# ------------------------------------------------------------------------------------------------------------------------
# N ranks from one command line
# ------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this command line as child processes, one per GPU
    (RANK = LOCAL_RANK = i), wait for them and pass rank 0's JSON line through.  Runs before this process imports torch or
    touches HIP — it never does either.  Returns the exit code (non-zero if any rank failed)."""
    def launch(port):
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None))
        return procs

    for attempt in range(3):  # the port is probed, released and reused: another process may take it in between
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = launch(port)
        # rank 0's output is drained by threads (its pipes must not fill up) while ALL children are polled: when one dies — before or at the
        # rendezvous, say — the others are ended instead of waiting for the process group's timeout
        chunks = {"out": [], "err": []}
        readers = [threading.Thread(target=lambda f, k: chunks[k].append(f.read()), args=(procs[0].stdout, "out"), daemon=True),
                   threading.Thread(target=lambda f, k: chunks[k].append(f.read()), args=(procs[0].stderr, "err"), daemon=True)]
        for t in readers:
            t.start()
        codes = [None] * n
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[r] = p.wait()
                break
            time.sleep(0.05)
        for t in readers:
            t.join(timeout=10)
        out0 = b"".join(chunks["out"]).decode("utf-8", "replace")
        err0 = b"".join(chunks["err"]).decode("utf-8", "replace")
        if any(c != 0 for c in codes) and ("EADDRINUSE" in err0 or "Address already in use" in err0 or "address already in use" in err0) and attempt < 2:
            print(f"[bench] rendezvous port {port} was taken, trying another", file=sys.stderr)
            continue
        sys.stderr.write(err0)
        sys.stdout.write(out0)
        sys.stdout.flush()
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad:
            print(f"[bench] ranks failed: {bad}", file=sys.stderr)
            return 1
        return 0
    return 1


# ------------------------------------------------------------------------------------------------------------------------
# workloads (host side, before the process touches the GPU: the scans are ray-cast on a pool of forked workers)
# ------------------------------------------------------------------------------------------------------------------------
def make_workload(n_distinct: int, batch: int, rank: int, prefilter_mode: str):
    """BASELINE config[1] shape. Returns (scene, poses, raw scans): n_distinct + 1 consecutive scans; pair k = (scan k, scan k + 1)."""
    from mrg_slam_amd import synth

    kitti = os.environ.get("KITTI_ROOT")
    if kitti and os.path.exists(os.path.join(kitti, "sequences", "00", "velodyne", "000000.bin")):
        # optional: KITTI odometry sequence 00, rank r starts at scan 500 r; lidar-frame ground truth
        # inv(Tr) * pose * Tr when poses/00.txt and calib.txt are there, else a 1 m/scan forward guess as "truth"
        first = 500 * rank
        scans = [synth.load_kitti_scan(first + k, kitti) for k in range(n_distinct + 1)]
        poses = None
        pf, cf = os.path.join(kitti, "poses", "00.txt"), os.path.join(kitti, "sequences", "00", "calib.txt")
        if os.path.exists(pf) and os.path.exists(cf):
            cam = np.loadtxt(pf)[first:first + n_distinct + 1].reshape(-1, 3, 4)
            tr = next(np.array(line.split()[1:], dtype=np.float64).reshape(3, 4) for line in open(cf) if line.startswith("Tr"))
            Tr = np.vstack([tr, [0, 0, 0, 1]])
            poses = [np.linalg.inv(Tr) @ np.vstack([c, [0, 0, 0, 1]]) @ Tr for c in cam]
        if poses is None:
            poses = [synth.make_pose([1.0 * k, 0.0, 0.0], np.eye(3)) for k in range(n_distinct + 1)]
        return None, poses, scans
    # every rank drives its own street (scene seed 1234 + rank), 1 m and +-1.5 deg of yaw per scan, weaving between the building rows
    scene = synth.street_scene(seed=1234 + rank, x_range=(-120.0, float(max(420, n_distinct + 140))))
    poses = synth.weave_trajectory(n_distinct + 1)
    seeds = [synth.BASE_SEED + 100000 * rank + k for k in range(n_distinct + 1)]
    scans = synth.synth_lidar_many(scene, poses, "VLP64", seeds, cache_tag=f"street_r{rank}_n{n_distinct + 1}")
    return scene, poses, scans


def make_loop_workload(n_keyframes: int = 64, n_pairs: int = 256, radius: float = 40.0, seed: int = 4242):
    """BASELINE config[3] (SURVEY.md §8d "C4"): keyframes on a ring road, (new keyframe, candidate) pairs with xy distance
    <= candidate_max_xy_distance = 15 m (config/mrg_slam.yaml:169), drawn with `seed`, guesses = relative pose of the graph
    estimates = truth perturbed by N(0, 0.5 m / 2 deg).  Same on every rank.  Returns (scans, pairs) with pairs sorted by new
    keyframe: (new keyframe index, candidate index, guess 4x4, true relative pose)."""
    from mrg_slam_amd import synth

    scene = synth.loop_scene(radius=radius)
    poses = synth.loop_trajectory(n_keyframes, radius)
    scans = synth.synth_lidar_many(scene, poses, "VLP64", [synth.BASE_SEED + 5000 + k for k in range(n_keyframes)], cache_tag=f"loop_k{n_keyframes}_r{int(radius)}")
    rng = np.random.default_rng(seed)
    cand = [(a, b) for a in range(n_keyframes) for b in range(n_keyframes)
            if a != b and np.hypot(*(poses[a][:2, 3] - poses[b][:2, 3])) <= 15.0]
    pick = sorted(rng.choice(len(cand), size=min(n_pairs, len(cand)), replace=False).tolist())
    pairs = []
    for i in pick:
        a, b = cand[i]
        rel = synth.rel_pose(poses[a], poses[b])  # candidate -> new keyframe frame (loop_detector.cpp:130)
        guess = synth.perturb_pose(rel, rng, sigma_t=(0.5, 0.5, 0.1), sigma_r_deg=(0.5, 0.5, 2.0))
        pairs.append((a, b, guess, rel))
    return scans, pairs


def latest_pmc_summary():
    """(name, dict) of the newest profiles/*_summary.json: the HBM traffic and VALU utilisation of the named kernels from separate rocprofv3 --pmc
    passes (profiles/collect*.sh, profiles/summarize.py), or (None, {}) when there is none."""
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.json"))
        if prof:
            return prof[-1], json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
    except (OSError, ValueError):
        pass
    return None, {}


def pmc_reference(name, pj):
    """Where the cached counter figures printed beside the live measurements come from (they are from another run and build)."""
    if not name:
        return None
    return {"profile": f"profiles/{name}", "git_rev": pj.get("git_rev"), "collected": pj.get("collected"),
            "note": "`traffic` / `valu_busy` are rocprofv3 --pmc figures of that profile's run (separate passes, collected at that revision), not of this run; "
                    "everything else in the record is measured live"}


def sha16(arrays) -> str:
    import hashlib

    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------------
# the driver's line
# ------------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 4096  # bytes; round 5's 20 KB line was not parsed by the driver (BENCH_r05.json: parsed null)
EXTRAS_FILE = "bench_extras.json"


def _sig(x, n=5):
    """floats to n significant digits (the line is a summary; the full-precision record is the extras file)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, (np.floating, np.integer)):
        return _sig(x.item(), n)
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _pick(d, *keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


def compact_line(full: dict) -> dict:
    """The ONE stdout line of a run: the contract's keys + `roofline` + `cpu_baseline` + the parity counts, under LINE_BUDGET bytes.  Everything else
    `full` holds (config[2] GICP, PCL NDT, the config[3] leg's phase times, notes) is written to EXTRAS_FILE and to stderr.
    `roofline` describes the dominant kernel ALONE on the chip: when the timed region keeps two batches in flight its launches overlap the other
    batch's and their HIP-event durations are those of two launches side by side — so the figure comes from the same K steps run one at a time
    just before the timed region (the shape profiles/*_rocprof_summary.md traces: `--in-flight 1`), and the timed region's own per-launch and
    aggregate figures ride beside it as `in_timed_region`."""
    g = full.get
    out = {k: g(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = g("config") or {}
    out["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload", ""))[:320]}
    for k in ("steps_in_flight", "pairs_per_gpu_per_step", "points_per_scan", "parallelism", "record_gather"):
        if cfg.get(k) is not None:
            out["config"][k] = cfg[k]
    roof = g("roofline")
    if roof:
        iso = roof.get("one_step_at_a_time") or None
        r = {"kernel": roof.get("kernel"), "bound": "hbm", "limited_by": roof.get("bound"), "peak": roof.get("peak"), "unit": roof.get("unit")}
        src = iso if iso else roof
        r.update({"achieved": src.get("achieved"), "frac": src.get("frac"), "avg_launch_ms": src.get("avg_launch_ms"), "launches": src.get("launches"),
                  "alg_bytes_per_launch": roof.get("alg_bytes_per_launch"), "traffic": roof.get("traffic"), "valu_busy": roof.get("valu_busy"),
                  "pmc_profile": roof.get("pmc_profile"),
                  "measured": "HIP events, the K steps one at a time before the timed region (kernel alone on the chip)" if iso else "HIP events over the timed region"})
        if iso:
            r["in_timed_region"] = {"avg_launch_ms": roof.get("avg_launch_ms"), "frac_per_overlapped_launch": roof.get("frac"),
                                    "aggregate_frac": roof.get("aggregate_frac_over_the_timed_region")}
        out["roofline"] = r
    else:
        out["roofline"] = None
    cpu = g("cpu_baseline")
    out["cpu_baseline"] = dict(_pick(cpu, "value", "unit", "cores", "kind"), sample=cpu.get("sample_short") or str(cpu.get("sample", ""))[:160]) if cpu else None
    par = g("parity_vs_oracle")
    out["parity_vs_oracle"] = _pick(par, "pairs", "pairs_over_bar", "max_dt_m", "max_dr_rad", "pairs_with_other_iterations_or_convergence") if par else None
    soak = g("soak_over_bar")
    if soak:
        def kn(keys):
            k = n = 0
            for key in keys:
                a, b = str(soak.get(key, "0/0")).split("/")
                k, n = k + int(a), n + int(b)
            return f"{k}/{n}"
        out["soak_over_bar"] = {"ndt": soak.get("ndt"), "ndt_reference_order": soak.get("ndt_reference_order"), "pcl_ndt": soak.get("pcl_ndt"),
                                "other": kn(("icp_gicp_vgicp_small_gicp", "pcl_gicp_serial", "pcl_gicp_omp", "icp_reciprocal")), "worst_m": soak.get("worst_m")}
    seq = g("value_one_step_at_a_time")
    if seq:
        out["value_one_step_at_a_time"] = seq.get("value")
    hp = g("value_host_pointers")
    if hp:
        pin = hp.get("pinned_host_clouds") or {}
        out["value_host_pointers"] = {"pageable": hp.get("value"), "page_locked": pin.get("value"),
                                      "page_locked_two_in_flight": (pin.get("two_batches_in_flight") or {}).get("value")}
    for k in ("single_pair_latency_ms", "evaluations_launched_per_alignment", "iterations_per_alignment", "mean_valid_neighbours", "converged"):
        if g(k) is not None:
            out[k] = g(k)
    ro = g("ndt_reference_order")
    if ro:
        out["ndt_reference_order"] = _pick(ro, "pairs", "ms_per_step", "ms_per_step_default_order")
    sp = g("gpu_split_ms_per_step")
    if sp:
        out["set_target_ms_per_step"] = sp.get("set_target_ms")
    c3 = g("config3_shard")
    if c3:
        ph = (c3.get("per_rank_phases_ms") or [{}])[0]
        p3 = c3.get("parity_vs_oracle") or {}
        out["config3"] = {"pairs": c3.get("pairs_total"), "ms_per_step": c3.get("ms_per_step"), "alignments_per_s": c3.get("alignments_per_s"),
                          "rounds_ms": ph.get("alignment_rounds"), "fitness_ms": ph.get("fitness_passes"), "build_ms": ph.get("build_targets"),
                          "two_in_flight_ms": (c3.get("two_batches_in_flight") or {}).get("ms_per_step"),
                          "fitness_frac": (c3.get("roofline_fitness") or {}).get("frac"), "derivative_frac": (c3.get("roofline") or {}).get("frac"),
                          "average_time_per_candidate_us": c3.get("average_time_per_candidate_us"),
                          "pairs_over_bar": p3.get("pairs_over_bar"), "best_candidate_mismatches": p3.get("best_candidate_mismatches"),
                          "records_sha256_16": c3.get("records_sha256_16")}
        db = c3.get("detect_batched")
        if db:
            out["config3"]["detect_8_new_keyframes"] = {g: _pick(db[g], "detect_ms", "detect_batched_ms", "alignments_sequential", "alignments_batched", "same_loops")
                                                        for g in ("no_gating", "default_gates") if g in db}
        if c3.get("shard_of_8_ms") is not None:
            out["config3"]["shard_of_8_ms"] = c3["shard_of_8_ms"]
    c2 = g("config2_gicp")
    if c2:
        out["config2_gicp"] = {}
        for m in ("GICP_HIP", "SMALL_GICP_HIP"):
            if isinstance(c2.get(m), dict):
                e = c2[m]
                out["config2_gicp"][m] = {"frame_ms": e.get("frame_ms"), "frame_ms_1m_from_keyframe": (e.get("frame_ms_by_displacement_m") or {}).get("1"),
                                          "keyframe_every_metre_ms": (e.get("keyframe_every_metre") or {}).get("frame_ms"),
                                          "batch32_ms": (e.get("batch_32_candidates") or {}).get("ms_per_call"),
                                          "frames_over_bar": (e.get("parity_vs_oracle") or {}).get("frames_over_bar"),
                                          "knn_frac": (e.get("roofline_knn") or {}).get("frac"), "linearize_frac": (e.get("roofline_linearize") or {}).get("frac")}
    for k in ("records_sha256_16", "raw_inputs_as_in_the_build_container"):
        if g(k) is not None:
            out[k] = g(k)
    out["extras"] = EXTRAS_FILE
    out = _sig(out)
    # the budget is a promise to the driver: shed the optional blocks, largest first, rather than print a line it cannot parse
    for k in ("config2_gicp", "config3", "soak_over_bar", "value_host_pointers"):
        if len(json.dumps(out)) < LINE_BUDGET:
            break
        out.pop(k, None)
    return out


def emit(full: dict, full_line: bool = False) -> None:
    """rank 0's output: the full record to stderr and to EXTRAS_FILE (gpurun_out/ when it exists, so that it travels back from the GPU box), then the
    compact line as the ONLY stdout line (`--full-line`: the full record on stdout instead, for the tools under profiles/ that read its side blocks)."""
    text = json.dumps(full)
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, EXTRAS_FILE), "w") as f:
                    f.write(text + "\n")
                break
            except OSError:
                pass
    if full_line:
        print(text)
    else:
        print("[bench extras] " + text, file=sys.stderr)
        line = json.dumps(compact_line(full))
        assert len(line) < LINE_BUDGET, len(line)
        print(line)
    sys.stdout.flush()


# digests of the RAW synthetic scans of the default workloads as generated in the build container (mrg_slam_amd/synth.py is built from IEEE
# + - * / sqrt alone, tests/test_synth_reproducible.py): the same value must come out on every host
EXPECTED_RAW_INPUTS = {"config1_rank0_257_scans": "656edc98f6ac7ec3", "config3_64_keyframes": "dc76f90d0d1b35a7"}


def run_config2(ctx, scans, dev, poses, lib, args):
    """BASELINE config[2]: scan-to-keyframe GICP on the ~130k-point scans (registrations.cpp:46-63: SMALL_GICP is the YAML default, FAST_GICP the
    code default), per frame setInputSource (k = 20 covariances) + align against a keyframe set once; kernel times from the library's HIP events."""
    from mrg_slam_amd import GicpHip, SmallGicpHip, synth
    from oracle import oracle as orc

    out = {"workload": f"keyframe = scan 0 ({len(scans[0])} points), frames = scans 1..6 (1 .. 6 m from the keyframe), guess = perturbed true motion (seed 5000+k), max_correspondence_distance 2.0, "
                       f"k = 20, eps {args.eps}, clouds resident in HBM"}
    frames = list(range(1, min(7, len(scans))))
    rels = {k: synth.rel_pose(poses[0], poses[k]) for k in frames}
    guesses = {k: synth.warm_guess(rels[k], 5000 + k) for k in frames}
    # loop-closure-sized perturbations of the same frames (0.5 m / 2 deg, seed 4242 + k: SURVEY.md §8d "C4"): alignments that need several outer iterations
    far_rng = {k: np.random.default_rng(4242 + k) for k in frames}
    far_guesses = {k: synth.perturb_pose(rels[k], far_rng[k], sigma_t=(0.5, 0.5, 0.1), sigma_r_deg=(0.5, 0.5, 2.0)) for k in frames}
    for name, cls, ocls in (("SMALL_GICP_HIP", SmallGicpHip, orc.SmallGicp), ("GICP_HIP", GicpHip, orc.FastGicp)):
        reg = cls(transformation_epsilon=args.eps, ctx=ctx)
        t_set, t_frame, lin, its, finals = [], [], np.zeros(3), [], {}
        knn = None
        for rep in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0]))
            reg.setInputSourceDevice(dev[frames[0]].data_ptr(), len(scans[frames[0]]))
            reg.align(guesses[frames[0]])  # the target's covariances and grid are built by the first align
            ctx.synchronize()
            t_set.append(1e3 * (time.perf_counter() - t0))
            for k in frames:
                t0 = time.perf_counter()
                reg.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
                reg.align(guesses[k])
                if rep:
                    t_frame.append(1e3 * (time.perf_counter() - t0))
                    lin += np.array(reg.kernel_stats())
                    its.append(reg.getFinalNumIteration())
                finals[k] = reg.getFinalTransformation()
            knn = ctx.knn_stats()
        # candidates the k-NN measures per query: one more frame with the diagnostic counters on
        lib().mrgfe_dbg_set_fit_stats(1)
        reg.setInputSourceDevice(dev[frames[0]].data_ptr(), len(scans[frames[0]]))
        reg.align(guesses[frames[0]])
        kc = ctx.knn_stats()
        lib().mrgfe_dbg_set_fit_stats(0)
        mbar = kc["candidates"] / kc["queries"] if kc["queries"] else 0.0
        knn_bytes = knn["queries"] * (16.0 + 27.0 * 8.0 + mbar * 16.0)
        knn_gbps = (knn_bytes / 1e9) / (knn["ms"] / 1e3) if knn["ms"] > 0 else 0.0
        lin_gbps = (lin[2] / 1e9) / (lin[0] / 1e3) if lin[0] > 0 else 0.0
        err = [float(np.linalg.norm(finals[k][:3, 3] - rels[k][:3, 3])) for k in frames]
        pmc_name, pmc = latest_pmc_summary()
        out["pmc_reference"] = pmc_reference(pmc_name, pmc)
        rec = {"first_frame_incl_setInputTarget_ms": float(np.median(t_set)), "frame_ms": float(np.median(t_frame)), "frames_timed": len(t_frame),
               "outer_iterations_per_frame": float(np.mean(its)), "median_translation_error_vs_truth_m": float(np.median(err)),
               "roofline_knn": {"bound": "latency", "byte_model_bound": "hbm", "kernel": "nn_knn_kernel (k = 20, one wavefront per query)", "achieved": knn_gbps, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": knn_gbps / HBM_PEAK_GBPS, "traffic": pmc.get("knn_traffic_bytes_per_launch"), "traffic_note": "PMC bytes per launch on a ~130k-point cloud "
                                "(profiles/gicp_profile.py batch), not on this frame's cloud", "valu_busy": pmc.get("knn_valu_busy"), "pmc_profile": pmc_name, "launch_ms": knn["ms"], "queries": knn["queries"],
                                "candidate_points_per_query": mbar, "byte_model": "N * (16 + 27*8 + m*16), m = candidates measured per query (counted)"},
               "roofline_linearize": {"bound": "latency", "byte_model_bound": "hbm", "kernel": "gicp_corr_kernel + gicp_linearize_kernel (one HIP-event pair around both)",
                                      "achieved": lin_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": lin_gbps / HBM_PEAK_GBPS, "traffic": None, "launches": int(lin[1]),
                                      "avg_launch_ms": lin[0] / lin[1] if lin[1] else None,
                                      "byte_model": "N_src * (16 + 48 + 27*8) + correspondences * (16 + 48) per linearisation (SURVEY.md §8d)"}}
        # frame time against the displacement from the keyframe (the correspondence search of a frame a few metres from its keyframe walks the occupancy pyramid
        # for the ground rings that fall between the keyframe's: profiles/gicp_corr_modes.py), and SURVEY.md §8(d)'s own shape of this config — a keyframe every
        # metre on the 1 m / scan trajectory: frame k against keyframe k - 1, the keyframe's setInputTarget (grid + k = 20 covariances) inside the time
        by_k = {k: [] for k in frames}
        for rep in range(3):
            for k in frames:
                ctx.synchronize()
                t0 = time.perf_counter()
                reg.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
                reg.align(guesses[k])
                by_k[k].append(1e3 * (time.perf_counter() - t0))
        rec["frame_ms_by_displacement_m"] = {str(k): float(np.median(v[1:])) for k, v in by_k.items()}
        kf = {}
        for mode in ("set_input_target", "source_becomes_target"):  # the keyframe handed over again / taken over from the source (mrgfe_reg_source_becomes_target: what the PCL adapter does)
            reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0]))
            t_kf, fin = [], []
            for rep in range(3):
                for k in frames:
                    g_k = synth.warm_guess(synth.rel_pose(poses[k - 1], poses[k]), 6000 + k)
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    reg.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
                    reg.align(g_k)
                    if mode == "set_input_target":
                        reg.setInputTargetDevice(dev[k].data_ptr(), len(scans[k]))  # (prepared by the next align: the steady loop's median holds it)
                    else:
                        reg.sourceBecomesTarget()
                    if rep:
                        t_kf.append(1e3 * (time.perf_counter() - t0))
                    fin.append(reg.getFinalTransformation().copy())
                reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0]))
            kf[mode] = (float(np.median(t_kf)), np.stack(fin))
        rec["keyframe_every_metre"] = {"frame_ms": kf["source_becomes_target"][0], "frame_ms_keyframe_handed_over_again": kf["set_input_target"][0], "frames_timed": len(t_kf),
                                       "same_transformations": bool(np.array_equal(kf["set_input_target"][1], kf["source_becomes_target"][1])),
                                       "note": "SURVEY.md §8(d) C3: frame k against keyframe k - 1 (1 m apart): setInputSource + align, then the frame becomes the keyframe "
                                               "(scan_matching_odometry_component.cpp:326-339) — taken over from the source with its covariances and grid, or handed over again"}
        reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0]))  # back to keyframe 0 for the legs below
        reg.setInputSourceDevice(dev[frames[0]].data_ptr(), len(scans[frames[0]]))
        reg.align(guesses[frames[0]])
        # the same frames from loop-closure-sized guesses: several outer iterations per alignment (the warm frames above converge in one)
        far = {}
        t_far = []
        for k in frames:
            ctx.synchronize()
            t0 = time.perf_counter()
            reg.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
            reg.align(far_guesses[k])
            t_far.append(1e3 * (time.perf_counter() - t0))
            far[k] = (reg.getFinalTransformation(), bool(reg.hasConverged()), int(reg.getFinalNumIteration()))
        rec["far_guess_frames"] = {"frame_ms": float(np.median(t_far)), "outer_iterations_per_frame": float(np.mean([v[2] for v in far.values()])),
                                   "guess": "true motion perturbed by N(0, 0.5 m / 2 deg), seed 4242 + k"}
        if not args.no_cpu:
            host_cores = os.cpu_count() or 1
            nt = min(32, host_cores)
            o = ocls(transformation_epsilon=args.eps, num_threads=nt)
            tc = time.perf_counter()
            o.setInputTarget(scans[0])
            o.setInputSource(scans[frames[0]])
            o.align(guesses[frames[0]])
            tc = time.perf_counter() - tc
            # parity: every warm frame and every far-guess frame against the oracle (the timing above is the one-frame sample)
            dts, drs, mism, its_o = [], [], 0, []
            for kind, gs, got in (("warm", guesses, {k: (finals[k], True, None) for k in frames}), ("far", far_guesses, far)):
                for k in frames:
                    o.setInputSource(scans[k])
                    o.align(gs[k])
                    To = o.getFinalTransformation()
                    Tg = got[k][0]
                    dts.append(float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])))
                    drs.append(rot_angle(Tg[:3, :3], To[:3, :3]))
                    if kind == "far":
                        mism += int(got[k][1] != bool(o.hasConverged()) or got[k][2] != int(o.getFinalNumIteration()))
                        its_o.append(int(o.getFinalNumIteration()))
            rec["cpu_oracle"] = {"ms": 1e3 * tc, "threads": nt, "sample": "one frame incl. setInputTarget (both clouds' k-NN covariances)", "kind": "port"}
            rec["parity_vs_oracle"] = {"frames": len(dts), "warm_frames": len(frames), "far_guess_frames": len(frames), "max_dt_m": max(dts), "max_dr_rad": max(drs),
                                       "frames_over_bar": int(sum(a > 1e-4 or b > 1e-4 for a, b in zip(dts, drs))), "frames_bit_identical": int(sum(a == 0.0 and b == 0.0 for a, b in zip(dts, drs))),
                                       "far_frames_with_other_iterations_or_convergence": mism, "far_frames_oracle_outer_iterations": its_o, "bar": "1e-4 m / 1e-4 rad"}
        out[name] = rec
    # the same method over a loop-closure-sized batch: 32 candidates (scans 1..4 in turn, warm guesses) against keyframe 0, k-NN covariances of every
    # candidate recomputed in the call (no keyframe store) - where the chip is full; the kernels behind it are priced in profiles/ (gicp_profile.py batch)
    from mrg_slam_amd import BatchMatcher
    from mrg_slam_amd._lib import SMALL_GICP_HIP
    from mrg_slam_amd.registration import default_params

    gp = default_params(SMALL_GICP_HIP)
    gp.transformation_epsilon = args.eps
    gb = BatchMatcher(gp, ctx)
    n_c = min(4, len(scans) - 1)
    cand = [1 + b % n_c for b in range(32)]
    g_args = ([dev[0].data_ptr()], [len(scans[0])], np.zeros(32, dtype=np.int32), [dev[k].data_ptr() for k in cand], [len(scans[k]) for k in cand],
              np.stack([synth.warm_guess(rels[k] if k in rels else synth.rel_pose(poses[0], poses[k]), 7000 + b) for b, k in enumerate(cand)]))
    t_b = []
    for rep in range(4):
        ctx.synchronize()
        t0 = time.perf_counter()
        gb.clear()
        gb.add_device(*g_args)
        rb = gb.align(-1.0)
        ctx.synchronize()
        t_b.append(1e3 * (time.perf_counter() - t0))
    out["SMALL_GICP_HIP"]["batch_32_candidates"] = {"ms_per_call": float(np.median(t_b[1:])), "ms_per_alignment": float(np.median(t_b[1:])) / 32.0, "converged": int(rb["converged"].sum()),
                                                    "points_per_cloud": int(np.mean([len(scans[k]) for k in cand])),
                                                    "note": "clear + add + align of 32 candidates against one keyframe, clouds resident, every candidate's k = 20 covariances recomputed in the call"}
    return out


def run_pcl_ndt(ctx, scans, dev, pairs, lib, args):
    """registration_method "NDT" = pcl::NormalDistributionsTransform (registrations.cpp:115-129) on BASELINE config[1]'s pairs: PCL_NDT_HIP, all pair terms
    in f64, radius search over the voxel centroids (27 probes per point).  Two epsilons: the run's own (mrg_slam's 0.1: PCL's iteration rule stops after
    ONE Newton step) and 1e-5 (tens of iterations).  Device time of the f64 derivative launches from the library's HIP events; byte model
    N * (16 + 27*8) + neighbours * 112 (f64 mean 24 + f64 inverse covariance 72 + float centroid 16)."""
    from mrg_slam_amd import BatchMatcher
    from mrg_slam_amd._lib import PCL_NDT_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    out = {"workload": f"{len(pairs)} distinct config[1] pairs, one setInputTarget per alignment, clouds resident in HBM, resolution 1.0, max_iterations 64"}
    add_args = ([dev[p[0]].data_ptr() for p in pairs], [len(scans[p[0]]) for p in pairs], np.arange(len(pairs), dtype=np.int32),
                [dev[p[1]].data_ptr() for p in pairs], [len(scans[p[1]]) for p in pairs], np.stack([p[2] for p in pairs]))
    for eps in (args.eps, 1e-5):
        prm = default_params(PCL_NDT_HIP)
        prm.transformation_epsilon, prm.maximum_iterations, prm.resolution = eps, 64, 1.0
        bm = BatchMatcher(prm, ctx)
        t, k = [], np.zeros(3)
        pts = nb = 0.0
        res = None
        for rep in range(4):
            ctx.synchronize()
            t0 = time.perf_counter()
            bm.clear()
            bm.add_device(*add_args)
            res = bm.align()
            ctx.synchronize()
            if rep:
                t.append(1e3 * (time.perf_counter() - t0))
                k += np.array(bm.kernel_stats())
                a, b = bm.pair_counts()
                pts, nb = pts + a, nb + b
        gbps = (k[2] / 1e9) / (k[0] / 1e3) if k[0] > 0 else 0.0
        rec = {"ms_per_step": float(np.median(t)), "alignments_per_s": len(pairs) / (np.median(t) / 1e3), "iterations_per_alignment": float(res["iterations"].mean()),
               "evaluations_per_alignment": float(res["evaluations"].mean()), "converged": int(res["converged"].sum()), "mean_neighbours_per_point": nb / pts if pts else 0.0,
               "median_translation_error_vs_truth_m": float(np.median([np.linalg.norm(result_matrix(res[b])[:3, 3] - pairs[b][3][:3, 3]) for b in range(len(pairs))])),
               "roofline": {"bound": "valu (f64)", "byte_model_bound": "hbm", "kernel": "ndt_derivatives_f64_all_kernel", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": gbps / HBM_PEAK_GBPS, "traffic": None, "launches": int(k[1]), "avg_launch_ms": k[0] / k[1] if k[1] else None,
                            "byte_model": "N * (16 + 27*8) + neighbours * 112 per evaluation"}}
        if not args.no_cpu:
            n_par = min(8, len(pairs))
            dts, drs, mism = [], [], 0
            tc = 0.0
            o = orc.PclNdt(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64)
            for b in range(n_par):
                ti, si, guess = pairs[b][0], pairs[b][1], pairs[b][2]
                t0 = time.perf_counter()
                o.setInputTarget(scans[ti])
                o.setInputSource(scans[si])
                o.align(guess)
                tc += time.perf_counter() - t0
                Tg, To = result_matrix(res[b]), o.getFinalTransformation()
                dts.append(float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])))
                drs.append(rot_angle(Tg[:3, :3], To[:3, :3]))
                mism += int(bool(res[b]["converged"]) != bool(o.hasConverged()) or int(res[b]["iterations"]) != int(o.getFinalNumIteration()) or int(res[b]["evaluations"]) != int(o.evals))
            rec["cpu_oracle"] = {"alignments_per_s": n_par / tc, "threads": 1, "kind": "port", "sample": f"{n_par} pairs incl. setInputTarget (pcl::NormalDistributionsTransform is single-threaded)"}
            rec["parity_vs_oracle"] = {"pairs": n_par, "max_dt_m": max(dts), "max_dr_rad": max(drs), "pairs_over_bar": int(sum(a > 1e-4 or b > 1e-4 for a, b in zip(dts, drs))),
                                       "pairs_with_other_iterations_evaluations_or_convergence": mism, "bar": "1e-4 m / 1e-4 rad"}
        out[f"eps_{eps:g}"] = rec
        del bm
    return out


def run_detect_leg(ctx, prm, loop_raw, radius=40.0, laps=4, n_new=8, reps=3):
    """LoopDetector::detect() (loop_detector.cpp:15-38) with SEVERAL new keyframes per call — the call shape the reference really has: 1-30 candidates per
    new keyframe, one matching() after the other.  `laps` earlier laps of the 64-keyframe ring are the graph (4 x 64 keyframes, robot "a"), `n_new`
    consecutive keyframes of the next lap arrive in one detect() call: ~30 candidates within 15 m each.  Timed: detect() — matching() per new keyframe,
    each a candidate batch + a consistency batch (the round-3 mirror) — against detect_batched() — ONE superset batch + one consistency batch + host
    replay.  Two gate settings: `no_gating` (accum_distance_thresh_same_robot 0: the sequential loop aligns every candidate, the two do the same
    alignments) and `default_gates` (15 m: a loop found for keyframe k prunes the same-robot candidates of the next three, so the sequential loop
    aligns fewer pairs than the superset).  Same Loop lists (asserted).  Untimed for `value`."""
    from mrg_slam_amd import BatchMatcher, distance_filter, synth
    from mrg_slam_amd.loop_detector import Edge, KeyFrame, LoopDetector, LoopManager

    clouds = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in loop_raw]
    n_ring = len(clouds)
    poses = synth.loop_trajectory(n_ring, radius)
    circ = 2.0 * np.pi * radius
    rng = np.random.default_rng(4243)
    kfs = []
    for lap in range(laps + 1):
        for k in range(n_ring):
            est = synth.perturb_pose(poses[k], rng, sigma_t=(0.3, 0.3, 0.05), sigma_r_deg=(0.3, 0.3, 1.0))
            kf = KeyFrame(id=len(kfs) + 1, cloud=clouds[k], estimate=est, accum_distance=lap * circ + circ * k / n_ring, slam_uuid="a", first_keyframe=(len(kfs) == 0))
            if kfs:
                prev = kfs[-1]
                rel = np.linalg.inv(kf.estimate) @ prev.estimate
                kf.prev_edge = Edge(kf, prev, rel)
                prev.next_edge = Edge(kf, prev, rel)
                kf.connected.add(prev.id)
                prev.connected.add(kf.id)
            kfs.append(kf)
    known, new = kfs[: laps * n_ring], kfs[laps * n_ring + 10: laps * n_ring + 10 + n_new]
    out = {}
    for name, thresh in (("no_gating", 0.0), ("default_gates", 15.0)):
        det = LoopDetector({"accum_distance_thresh_same_robot": thresh}, matcher=BatchMatcher(prm, ctx))
        t_seq, t_bat, loops = [], [], {}
        for rep in range(reps + 1):  # the first repetition uploads the candidates' clouds into the keyframe store (keys): untimed
            for mode in ("sequential", "batched"):
                det.loop_manager = LoopManager()
                a0 = det.alignments
                ctx.synchronize()
                t0 = time.perf_counter()
                found = det.detect_batched(known, new) if mode == "batched" else det.detect(known, new)
                dt = 1e3 * (time.perf_counter() - t0)
                loops[mode] = [(lp.key1.id, lp.key2.id, lp.relative_pose.tobytes()) for lp in found]
                if rep:
                    (t_bat if mode == "batched" else t_seq).append(dt)
                if mode == "batched":
                    info = dict(det.last_batched)
                else:
                    seq_aligned = det.alignments - a0
        out[name] = {"detect_ms": float(np.median(t_seq)), "detect_batched_ms": float(np.median(t_bat)), "speedup": float(np.median(t_seq) / np.median(t_bat)),
                     "new_keyframes": n_new, "alignments_sequential": int(seq_aligned), "alignments_batched": int(info["superset_pairs"] + info["consistency_pairs"]),
                     "superset_pairs": info["superset_pairs"], "loops": len(loops["batched"]), "same_loops": loops["batched"] == loops["sequential"]}
        del det
    out["what"] = (f"{n_new} new keyframes per detect() call against {laps} x {n_ring} ring keyframes of one robot (~{out['no_gating']['superset_pairs'] // n_new} candidates within 15 m each, "
                   f"{int(np.mean([len(c) for c in clouds]))} points per cloud, candidates resident in the HBM keyframe store), NDT_HIP DIRECT7 res 1.0, median of {reps} calls")
    return out


def run_inproc(args, loop_raw, loop_pairs):
    """`--mode shard --inproc --gpus N`: BASELINE config[3] through mrgfe_node_* (csrc/node.cpp) — ONE process, one member (context + batch + host
    thread) per GPU, contiguous blocks of the keyframe-ordered pair list, the 384-byte records gathered by one ncclAllGather between distinct devices
    (host memory when members share a card: a one-GPU box rehearses N members on device 0), the best-candidate rule replayed on the gathered records
    (mrgfe_node_select_best).  What the reference's single host process per robot can call.  Prints the same record digest as the
    one-rank `--mode shard` run."""
    import hashlib

    import torch  # noqa: F401  (before libmrgfe)

    from mrg_slam_amd import NodeMatcher, distance_filter
    from mrg_slam_amd._lib import NDT_HIP, SEARCH
    from mrg_slam_amd.registration import default_params

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    devices = [g % ndev for g in range(args.gpus)]
    prm = default_params(NDT_HIP)
    prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method, prm.num_threads = args.eps, 64, 1.0, SEARCH["DIRECT7"], 8
    l_host = [distance_filter(s, 0.1, 35.0) for s in loop_raw]
    node = NodeMatcher(devices, prm)
    group_first = [0] + [i for i in range(1, len(loop_pairs)) if loop_pairs[i][0] != loop_pairs[i - 1][0]] + [len(loop_pairs)]

    def step(first=False):
        node.clear()
        tid = {}
        for a, b, guess, _ in loop_pairs:
            if a not in tid:  # registration_->setInputTarget(new_keyframe->cloud), once per new keyframe (loop_detector.cpp:104)
                tid[a] = node.add_target(l_host[a] if first else None, key=100000 + a, n_points=len(l_host[a]))
            node.add_pair(tid[a], l_host[b] if first else None, guess, key=1 + b, n_points=len(l_host[b]))
        res = node.align(float("inf"))
        return res, NodeMatcher.select_best(res, group_first)

    step(first=True)  # the clouds go up once and stay resident on the members that use them (keys): inputs in HBM when the timed region starts
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        res, best = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    elapsed = time.perf_counter() - t0
    digest = hashlib.sha256(res["T"].tobytes() + res["fitness"].tobytes() + res["converged"].tobytes()).hexdigest()[:16]
    in_digest = hashlib.sha256(b"".join(np.ascontiguousarray(c).tobytes() for c in l_host)).hexdigest()[:16]
    err = [float(np.linalg.norm(np.asarray(res[i]["T"]).reshape(4, 4).T[:3, 3] - loop_pairs[i][3][:3, 3])) for i in range(len(loop_pairs))]
    print(f"[bench inproc] per-step ms: " + " ".join(f"{v:.2f}" for v in step_ms), file=sys.stderr)
    emit({
        "metric": "scan-pair alignments/sec (NDT, ~120k pts, 1.0 m voxel) at 1/2/4/8 MI355X; HBM GB/s achieved", "value": len(loop_pairs) * args.steps / elapsed,
        "unit": "alignments/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32 per-pair terms, f64 accumulation", "data": "synthetic",
        "config": {"workload": f"BASELINE config[3] through mrgfe_node_* in ONE process: {len(loop_pairs)} loop-closure candidate pairs over {len(l_host)} keyframes, "
                               f"{args.gpus} members on devices {devices}, contiguous blocks, NDT_HIP DIRECT7 res 1.0 eps {args.eps}, getFitnessScore(inf), "
                               "keyed clouds resident in HBM, record gather + best-candidate replay inside the step",
                   "pairs_per_step": len(loop_pairs), "members": args.gpus, "devices": devices, "record_gather": node.last_gather(),
                   "blocks": [node.shard(m) for m in range(args.gpus)]},
        "records_sha256_16": digest, "inputs_sha256_16": in_digest, "converged": int(res["converged"].sum()),
        "loops_found": int(sum(b is not None for b, _ in best)), "median_translation_error_vs_truth_m": float(np.median(err)),
        "mean_points_per_scan": float(np.mean([len(c) for c in l_host])),
        "roofline": None, "cpu_baseline": None,
        "note": "side mode (never the driver's line): the node path behind the C ABI; compare records_sha256_16 with `--mode shard` on one rank"}, args.full_line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["weak", "shard"], default="weak", help="weak: B own pairs per rank (config[1] shape); shard: config[3], 256 pairs over all ranks")
    ap.add_argument("--batch", type=int, default=256, help="scan pairs per step and per GPU (more pairs in flight keep the GPU full in the late rounds)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic scan pairs generated per rank (0: = --batch, every pair its own two scans)")
    ap.add_argument("--cold-every", type=int, default=4, help="every n-th pair starts from the identity guess (0: warm guesses only)")
    ap.add_argument("--prefilter", choices=["distance", "full"], default="distance")
    ap.add_argument("--eps", type=float, default=0.1, help="reg_transformation_epsilon (config/mrg_slam.yaml:102)")
    ap.add_argument("--cpu-pairs", type=int, default=64, help="pairs of the bounded CPU-oracle sample, ~10-20 s of CPU work (0 disables)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--shard-steps", type=int, default=3, help="steps of the config[3] side measurement of a --mode weak run (0 disables)")
    ap.add_argument("--shard-of", type=int, default=0, help="--mode shard on ONE GPU: run only rank 0's shard of an N-rank job (no process group): what each GPU of an "
                                                           "N-GPU node would do per step, for projecting the strong scaling where N GPUs are not at hand")
    ap.add_argument("--inproc", action="store_true", help="--mode shard through mrgfe_node_* (csrc/node.cpp): ONE process, --gpus members (on distinct devices where the box has "
                                                          "them, sharing device 0 otherwise), no process group; prints the record digest of the one-rank run")
    ap.add_argument("--no-seq", action="store_true", help="skip the one-step-at-a-time pass in front of a pipelined timed region (profiling runs: every derivative launch of the "
                                                         "command then belongs to the pipelined steps)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight in the timed region (mrgfe_batch_align_async / _wait on as many contexts); 1: one step at a time")
    ap.add_argument("--prepare-only", action="store_true", help="generate (and cache) the synthetic scans, then exit without touching the GPU")
    ap.add_argument("--parity-pairs", type=int, default=0, help="pairs of the step checked against the CPU oracle (0: all of them; the CPU TIMING uses --cpu-pairs)")
    ap.add_argument("--no-shard-parity", action="store_true", help="skip the 256-pair oracle loop of the config[3] leg (parity_vs_oracle of config3_shard)")
    ap.add_argument("--soak-cases", type=int, default=120, help="random small scenes of the parity soak printed as soak_over_bar (0 disables; pcl::GICP / reciprocal ICP get a third as many)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs behind the main line (setInputTarget / align split, host-pointer rate, pipeline shape, config[2] GICP)")
    ap.add_argument("--full-line", action="store_true", help="print the FULL record as the stdout line (tools under profiles/ and tests that read its side blocks); default: "
                                                              "the compact line (< 4 KB) on stdout, the full record on stderr and in bench_extras.json")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-pair setInputTarget + align latency behind the timed region (profiling runs: it adds differently "
                                                               "sized launches of the same kernels, so the trace averages would no longer describe the timed workload)")
    args = ap.parse_args()
    if args.distinct <= 0:
        args.distinct = args.batch

    if args.inproc:
        if args.mode != "shard":
            raise SystemExit("--inproc drives config[3]: use it with --mode shard")
        loop_raw, loop_pairs = make_loop_workload()
        return run_inproc(args, loop_raw, loop_pairs)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the launcher never touches the GPU: it ray-casts and caches EVERY rank's scans first, one rank after the other with the whole host to
        # itself, so that the children load them instead of N ray-caster pools sharing the host cores behind the rendezvous
        if not os.environ.get("BENCH_SPAWN_TEST"):
            t_all = time.time()
            if args.mode == "weak":
                for r in range(args.gpus):
                    make_workload(args.distinct, args.batch, r, args.prefilter)
            if args.mode == "shard" or args.shard_steps > 0:
                make_loop_workload()
            print(f"[bench] scans of {args.gpus} ranks ready in {time.time() - t_all:.1f} s", file=sys.stderr)
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_SPAWN_TEST"):  # CPU test hook of the launcher above: report the rank environment and stop before any GPU work
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rank": rank, "local_rank": local_rank, "master": os.environ.get("MASTER_ADDR"), "port": os.environ.get("MASTER_PORT")}))
        if int(os.environ.get("BENCH_SPAWN_TEST_HANG_RANK", "-1")) == rank:
            time.sleep(120)  # stands for a rank waiting at a rendezvous that will never complete
        sys.exit(int(os.environ.get("BENCH_SPAWN_TEST_FAIL_RANK", "-1")) == rank)

    # ---- inputs, host part (untimed; forks worker processes, so it runs before the GPU is initialised) ------------------
    from mrg_slam_amd import synth

    t_gen = time.time()
    scene = poses = raw = None
    if args.mode == "weak":
        scene, poses, raw = make_workload(args.distinct, args.batch, rank, args.prefilter)
    loop_raw = loop_pairs = None
    if args.mode == "shard" or args.shard_steps > 0:
        loop_raw, loop_pairs = make_loop_workload()
    if args.prepare_only:
        print(f"[bench rank {rank}] synthetic scans ready in {time.time() - t_gen:.1f} s", file=sys.stderr)
        return

    import torch  # before libmrgfe: it then binds to the HIP runtime already in the process
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()  # test hook only: gloo ranks may share a GPU (one rank per GPU otherwise)
    torch.cuda.set_device(local_rank)
    # a launcher (torchrun / bench.py's own) sets WORLD_SIZE: then the process group exists and the records are gathered even when the job has ONE rank
    # (`torchrun --nproc-per-node 1`: RCCL initialisation and all_gather_into_tensor on device tensors run on a one-GPU box, tests/test_gpu_multiprocess.py)
    use_pg = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; BENCH_DIST_BACKEND=gloo lets two ranks share one GPU to exercise this path on a 1-GPU box (RCCL
        # refuses duplicate devices)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    gdev = torch.device("cuda", local_rank)

    from mrg_slam_amd import BatchMatcher, Context, NdtHip, distance_filter, prefilter
    from mrg_slam_amd import loop_closure as lc
    from mrg_slam_amd._lib import NDT_HIP, SEARCH, lib
    from mrg_slam_amd.registration import RESULT_DTYPE, default_params, result_matrix

    ctx = Context(local_rank)

    def to_hbm(raw_scans):
        host = [prefilter(s, ctx=ctx) if args.prefilter == "full" else distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw_scans]
        return host, [torch.from_numpy(s).to(gdev) for s in host]

    prm = default_params(NDT_HIP)
    prm.transformation_epsilon = args.eps
    prm.maximum_iterations = 64
    prm.resolution = 1.0
    prm.nn_search_method = SEARCH["DIRECT7"]
    prm.num_threads = 8
    bm = BatchMatcher(prm, ctx)

    def all_gather_records(local: np.ndarray, per: int):
        """the pose / Hessian record gather of the north star: `per` 384-byte records per rank over RCCL (gloo in the test hook)"""
        pad = np.zeros(per, dtype=RESULT_DTYPE)
        pad["pair_id"] = -1
        pad[: len(local)] = local
        mine = torch.from_numpy(pad.view(np.uint8).reshape(per, -1))
        if backend == "nccl":
            mine = mine.to(gdev)
            out = torch.empty((world * per, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=gdev)
        else:
            out = torch.empty((world * per, RESULT_DTYPE.itemsize), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, mine)
        rec = np.frombuffer(out.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
        return rec[rec["pair_id"] >= 0]

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()

    def timed(step_fn, steps, warmup, drain_fn=None):
        """drain_fn: steps that are submitted and collected later (--in-flight > 1) — whatever is still in flight is collected INSIDE the timed
        region, before the closing synchronisation: K steps are submitted and K steps complete between the two clock reads"""
        for _ in range(warmup):
            step_fn()
        if drain_fn and warmup:
            drain_fn()
        sync()
        # a full (generation 2) collection walks every object `import torch` created: ~40 ms, once, at an arbitrary step.
        # Collect now and move the survivors to the permanent generation so the timed steps are not interrupted by it.
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        step_ms, last = [], None
        for _ in range(steps):
            ts = time.perf_counter()
            r = step_fn()
            last = r if r is not None else last
            step_ms.append(1e3 * (time.perf_counter() - ts))
        if drain_fn:
            r = drain_fn()
            last = r if r is not None else last
        sync()
        elapsed = time.perf_counter() - t0
        if use_pg:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=gdev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, step_ms, last

    # ---- config[3]: 256 loop-closure candidate pairs sharded over the ranks ------------------------------------------------
    def run_shard(steps, warmup):
        l_host, l_dev = to_hbm(loop_raw)
        n_pairs = len(loop_pairs)
        fake_world = args.shard_of if (args.shard_of > 1 and world == 1) else 0
        mine = lc.shard_indices(n_pairs, fake_world or world, rank)
        per = -(-n_pairs // world)
        my_targets = sorted({loop_pairs[i][0] for i in mine})  # the new keyframes this rank needs a grid for (loop_detector.cpp:104)
        groups = {}
        for i, (a, _, _, _) in enumerate(loop_pairs):
            groups.setdefault(a, []).append(i)

        group_keys, group_ids = list(groups.keys()), list(groups.values())
        tpos = {a: k for k, a in enumerate(my_targets)}
        shard_args = ([l_dev[a].data_ptr() for a in my_targets], [len(l_host[a]) for a in my_targets], np.array([tpos[loop_pairs[i][0]] for i in mine], dtype=np.int32),
                      [l_dev[loop_pairs[i][1]].data_ptr() for i in mine], [len(l_host[loop_pairs[i][1]]) for i in mine],
                      np.stack([loop_pairs[i][2] for i in mine]) if len(mine) else np.zeros((0, 4, 4)))

        phases = [] if os.environ.get("BENCH_STEP_PHASES") else None  # diagnostic: host-side split of a step (queue the batch / align / records)

        def step():
            t0 = time.perf_counter()
            bm.clear()
            if len(mine):
                bm.add_device(*shard_args)
            t1 = time.perf_counter()
            local = bm.align(float("inf"))  # getFitnessScore(fitness_score_max_range = .inf), config/mrg_slam.yaml:172
            t2 = time.perf_counter()
            local["pair_id"] = mine.astype(np.int32)
            rec = all_gather_records(local, per) if use_pg else local
            full = np.zeros(n_pairs, dtype=RESULT_DTYPE)
            full["pair_id"] = -1
            full[rec["pair_id"]] = rec
            # every rank replays the sequential best-candidate rule per new keyframe (loop_detector.cpp:126-145)
            best = dict(zip(group_keys, lc.select_best_groups(full, group_ids)))  # (with --shard-of the other ranks' records are missing: unconverged zeros)
            if phases is not None:
                phases.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
            return full, best

        deriv = np.zeros(3)   # (device ms, launches, algorithmic bytes) of the derivative launches, HIP events around every launch
        fit_acc = {"ms_block": 0.0, "ms_sweep": 0.0, "ms_far": 0.0, "queued": 0.0, "queries": 0.0, "queued_far": 0.0, "launches": 0.0}
        acc_on = {"on": False}
        kind_bytes, kind_points, s_largest = np.zeros(3), np.zeros(3), {"ms": 0.0, "pairs": [0, 0, 0]}
        inner_step = step

        def step():  # noqa: F811 - the timed step plus the library's own accounting of it (read back after the step: host-side only)
            out = inner_step()
            if acc_on["on"]:
                deriv[:] += np.array(bm.kernel_stats(-1))
                for m in range(3):
                    kind_bytes[m] += bm.kernel_stats(m)[2]
                    kind_points[m] += bm.pair_counts(m)[0]
                lm, lp = bm.largest_launch()
                if lm > s_largest["ms"]:
                    s_largest["ms"], s_largest["pairs"] = lm, lp
                fs = bm.fitness_stats()
                for k in fit_acc:
                    fit_acc[k] += fs[k]
            return out

        bm.timing(reset=True)
        elapsed, step_ms, (full, best) = timed(step, steps, warmup)
        us_per_candidate = bm.timing()["average_time_per_candidate_us"]  # the reference's own statistic (apps/mrg_slam_component.cpp:1032-1037) over the timed steps + warm-up
        if phases:
            print("[bench] step phases, median ms: queue the batch %.3f, align %.3f, records + best candidate %.3f" % tuple(1e3 * np.median(np.array(phases[-steps:]), axis=0)), file=sys.stderr)
        # kernel times for the roofline records: two more, untimed, steps WITHOUT the early fitness waves (beside the alignment rounds the
        # kernels of both share the chip and their HIP-event times say little about either), then one with the sweep's counters on:
        # candidate points measured per queued query (the m-bar of SURVEY.md §8d)
        acc_steps = 2
        os.environ["MRGFE_NO_EARLY_FIT"] = "1"
        step()
        acc_on["on"] = True
        for _ in range(acc_steps):
            step()
        acc_on["on"] = False
        lib().mrgfe_dbg_set_fit_stats(1)
        inner_step()
        cs = bm.fitness_stats()
        lib().mrgfe_dbg_set_fit_stats(0)
        del os.environ["MRGFE_NO_EARLY_FIT"]
        mbar = cs["points"] / cs["queued"] if cs["queued"] else 0.0
        # per-rank phase times (host clock, a synchronisation between the phases: NOT how the timed step runs — there the phases follow each other
        # without a host wait — but what each rank spends where, so that a measured scaling curve explains its own efficiency)
        NPH = 7
        ph = np.zeros((3, NPH))
        for rep in range(ph.shape[0]):
            ctx.synchronize()
            t0 = time.perf_counter()
            bm.clear()
            if len(mine):
                bm.add_device(*shard_args)
            t1 = time.perf_counter()
            bm.build_targets()
            ctx.synchronize()
            t2 = time.perf_counter()
            bm.align(-1.0)  # the alignment rounds alone (no getFitnessScore)
            t3 = time.perf_counter()
            bm.clear()
            if len(mine):
                bm.add_device(*shard_args)
            ctx.synchronize()
            t4 = time.perf_counter()
            local = bm.align(float("inf"))
            t5 = time.perf_counter()
            local["pair_id"] = mine.astype(np.int32)
            rec = all_gather_records(local, per) if use_pg else local
            t6 = time.perf_counter()
            fl = np.zeros(n_pairs, dtype=RESULT_DTYPE)
            fl["pair_id"] = -1
            fl[rec["pair_id"]] = rec
            lc.select_best_groups(fl, group_ids)
            t7 = time.perf_counter()
            ph[rep] = [t1 - t0, t2 - t1, t3 - t2, (t5 - t4) - (t3 - t1), t6 - t5, t7 - t6, t5 - t4]
        ph_med = 1e3 * np.median(ph, axis=0)
        if use_pg:
            tt = torch.from_numpy(ph_med.copy())
            tt = tt.to(gdev) if backend == "nccl" else tt
            allp = torch.empty(world * NPH, dtype=torch.float64, device=tt.device)
            dist.all_gather_into_tensor(allp, tt)
            allp = allp.cpu().numpy().reshape(world, NPH)
        else:
            allp = ph_med.reshape(1, NPH)
        # Two batches in flight (mrgfe_batch_align_async on two contexts): the loop-closure batches of two new keyframe sets — two robots of the
        # multi-robot system, or two consecutive keyframe updates — submitted back to back; one batch's straggler rounds and fitness tail are filled
        # by the other's launches.  Untimed for `value` / `ms_per_step`; same records per batch (asserted).
        pipe = None
        if world == 1 and len(mine) and not os.environ.get("BENCH_NO_SHARD_PIPE"):
            bms2 = [bm, BatchMatcher(prm, Context(local_rank))]
            n_pipe = max(4, steps)

            def submit(b):
                b.clear()
                b.add_device(*shard_args)
                b.align_async(float("inf"))

            submit(bms2[1])
            ref_rec = bms2[1].wait()
            ctx.synchronize()
            tq = time.perf_counter()
            live = [False, False]
            same = True
            for it in range(n_pipe):
                k = it % 2
                if live[k]:
                    same = same and bool(np.array_equal(bms2[k].wait()["T"], ref_rec["T"]))
                submit(bms2[k])
                live[k] = True
            for j in range(2):
                k = (n_pipe + j) % 2
                if live[k]:
                    r2 = bms2[k].wait()
                    same = same and bool(np.array_equal(r2["T"], ref_rec["T"]) and np.array_equal(r2["fitness"], ref_rec["fitness"]))
            tq = time.perf_counter() - tq
            pipe = {"ms_per_step": 1e3 * tq / n_pipe, "steps": n_pipe, "steps_in_flight": 2, "alignments_per_s": len(mine) * n_pipe / tq, "same_records_every_step": same,
                    "note": "two loop-closure batches in flight on two contexts (mrgfe_batch_align_async / _wait); throughput, not the latency of one batch"}
            del bms2
        phase_names = ("queue_the_batch", "build_targets", "alignment_rounds", "fitness_passes", "record_gather", "best_candidate_replay", "align_call_with_fitness")
        per_rank_phases = [dict({"rank": r}, **{k: round(float(v), 3) for k, v in zip(phase_names, allp[r])}) for r in range(world)]
        have = [i for i in range(n_pairs) if not fake_world or i in set(mine.tolist())]
        err = [float(np.linalg.norm(result_matrix(full[i])[:3, 3] - loop_pairs[i][3][:3, 3])) for i in have]
        digest = __import__("hashlib").sha256(full["T"].tobytes() + full["fitness"].tobytes() + full["converged"].tobytes()).hexdigest()[:16]
        # the records are a function of the inputs: the synthetic scans are ray-cast with numpy on the host (its SIMD paths differ from CPU to CPU
        # in the last bit), so the digest of the inputs goes with the digest of the records
        in_digest = __import__("hashlib").sha256(b"".join(np.ascontiguousarray(c).tobytes() for c in l_host)).hexdigest()[:16]
        d_gbps = (deriv[2] / 1e9) / (deriv[0] / 1e3) if deriv[0] > 0 else 0.0
        # getFitnessScore's seed + sweep pass (nn_fit_seed_kernel + nn_fit_sweep_kernel, HIP events around the two): SURVEY.md §8(d) prices a 1-NN
        # query on the hash grid at 16 + 27 * 8 + m-bar * 16 bytes; N = the queries the 3x3x3 block did not settle
        f_bytes = fit_acc["queued"] * (16.0 + 27.0 * 8.0 + mbar * 16.0)
        f_gbps = (f_bytes / 1e9) / (fit_acc["ms_sweep"] / 1e3) if fit_acc["ms_sweep"] > 0 else 0.0
        pmc_name, pmc = latest_pmc_summary()
        sh_pmc = pmc.get("shard_pmc", {})
        d_pmc = sh_pmc.get("kernels", {}).get("ndt_derivatives_all_kernel<7>")
        # (the counter passes profile the whole 256-pair step on one GPU: per-launch / per-step traffic is quoted for that shape only)
        full_shape = len(mine) == n_pairs
        roof = {"bound": "valu", "byte_model_bound": "hbm", "kernel": "ndt_derivatives_all_kernel<7>", "achieved": d_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": d_gbps / HBM_PEAK_GBPS, "traffic": (d_pmc["hbm_bytes"] / d_pmc["launches"]) if (d_pmc and full_shape) else None,
                "valu_busy": d_pmc["valu_busy"] if d_pmc else None, "pmc_profile": pmc_name, "launches": int(deriv[1]), "avg_launch_ms": deriv[0] / deriv[1] if deriv[1] else None,
                "alg_bytes_per_launch": deriv[2] / deriv[1] if deriv[1] else None, "ms_per_step": deriv[0] / acc_steps,
                "byte_model": "per launch: sum over the evaluations of all active pairs of N_src*(16 + 7*8) + valid_neighbours*48 (SURVEY.md §8d); PMC traffic of this "
                              "kernel: profiles/r03_rocprof_summary.md"}
        n_src_mean = float(np.mean([len(l_host[loop_pairs[i][1]]) for i in mine])) if len(mine) else 0.0
        if s_largest["ms"] > 0 and n_src_mean > 0:
            per_eval = [kind_bytes[m] / (kind_points[m] / n_src_mean) if kind_points[m] > 0 else 0.0 for m in range(3)]
            lb = float(sum(n * b for n, b in zip(s_largest["pairs"], per_eval)))
            roof["largest_launch"] = {"ms": s_largest["ms"], "busy_pairs_by_kind": s_largest["pairs"], "alg_bytes": lb, "achieved_GBps": lb / 1e9 / (s_largest["ms"] / 1e3),
                                      "frac": lb / 1e9 / (s_largest["ms"] / 1e3) / HBM_PEAK_GBPS}
        roof_fit = {"bound": "valu", "byte_model_bound": "hbm", "kernel": "nn_fit_seed_kernel + nn_fit_sweep_kernel (getFitnessScore(inf), the queries their 3x3x3 block does not settle)",
                    "achieved": f_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": f_gbps / HBM_PEAK_GBPS,
                    "traffic": pmc.get("fitness_sweep_traffic_bytes_per_step") if full_shape else None, "traffic_unit": "HBM bytes per step (compare alg_bytes_per_step)",
                    "valu_busy": pmc.get("fitness_sweep_valu_busy"), "pmc_profile": pmc_name,
                    "queued_queries_per_step": fit_acc["queued"] / acc_steps, "queries_per_step": fit_acc["queries"] / acc_steps, "unseeded_queries_per_step": fit_acc["queued_far"] / acc_steps,
                    "candidate_points_per_queued_query": mbar, "alg_bytes_per_step": f_bytes / acc_steps, "ms_per_step": fit_acc["ms_sweep"] / acc_steps,
                    "block_pass_ms_per_step": fit_acc["ms_block"] / acc_steps, "pyramid_walk_ms_per_step": fit_acc["ms_far"] / acc_steps,
                    "byte_model": "N_queued * (16 + 27*8 + m*16), m = candidate points measured per queued query (counted in one extra untimed step); kernel times from "
                                  f"{acc_steps} untimed steps without the early fitness waves (MRGFE_NO_EARLY_FIT=1: one fitness launch per step, behind the alignment)"}
        # parity at the stated size: every pair of the step against the oracle running the reference's sequential loop per new keyframe
        # (loop_detector.cpp:104,126-145: transform, convergence, iterations, getFitnessScore(inf), best candidate); over-the-bar pairs are replayed in
        # the kernels' summation order (oracle/replay.py) — rank 0 of the 1-GPU run only, outside every timed region
        parity = None
        if not args.no_cpu and rank == 0 and world == 1 and full_shape and not args.no_shard_parity:
            from oracle.replay import loop_parity

            def single(i):
                reg = NdtHip(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, ctx=ctx)
                a, b = loop_pairs[i][0], loop_pairs[i][1]
                reg.setInputTargetDevice(l_dev[a].data_ptr(), len(l_host[a]))
                reg.setInputSourceDevice(l_dev[b].data_ptr(), len(l_host[b]))
                reg.align(loop_pairs[i][2])
                return reg.getFinalTransformation(), reg.hasConverged(), reg.getFinalNumIteration()

            tp = time.perf_counter()
            parity = loop_parity(l_host, loop_pairs, full, args.eps, single_runner=single)
            parity["seconds"] = time.perf_counter() - tp
        raw_digest = sha16(loop_raw)
        return {"pairs_total": n_pairs, "new_keyframes": len(groups), "pairs_per_gpu": int(len(mine)), "targets_built_per_gpu": len(my_targets),
                "parity_vs_oracle": parity, "raw_inputs_sha256_16": raw_digest,
                "two_batches_in_flight": pipe,
                "per_rank_phases_ms": per_rank_phases,
                "per_rank_phases_note": "median of 3 untimed steps with a host synchronisation between the phases (build_targets | align without fitness | the full align call again, "
                                        "fitness_passes = that call minus build and rounds | all-gather of the 384-byte records | best-candidate replay); the timed step runs them back to back",
                "raw_inputs_as_in_the_build_container": (raw_digest == EXPECTED_RAW_INPUTS["config3_64_keyframes"]) if EXPECTED_RAW_INPUTS["config3_64_keyframes"] else None,
                "pmc_reference": pmc_reference(pmc_name, pmc),
                "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "alignments_per_s": n_pairs * steps / elapsed, "scaling": "strong",
                "average_time_per_candidate_us": us_per_candidate,
                "projected_for_gpus": fake_world or None,
                "fitness_max_range": "inf", "converged": int(full["converged"].sum()), "matched_keyframes": int(sum(b[0] is not None for b in best.values())),
                "median_translation_error_vs_truth_m": float(np.median(err)), "mean_iterations": float(full["iterations"][have].mean()),
                "mean_points_per_scan": float(np.mean([len(s) for s in l_host])), "records_sha256_16": digest, "inputs_sha256_16": in_digest,
                "per_step_ms": [round(v, 2) for v in step_ms], "roofline": roof, "roofline_fitness": roof_fit}

    if args.mode == "shard":
        r = run_shard(args.steps, args.warmup)
        if rank == 0:
            out = {"metric": METRIC, "value": r["alignments_per_s"], "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 per-pair terms, f64 accumulation",
                   "data": "synthetic",
                   "config": {"workload": f"BASELINE config[3]: {r['pairs_total']} loop-closure candidate pairs ({r['new_keyframes']} new keyframes on a 40 m ring, VLP-64, "
                                          f"mean {r['mean_points_per_scan']:.0f} pts/scan), NDT_HIP DIRECT7 res 1.0 eps {args.eps}, contiguous blocks of the keyframe-ordered pair list per rank, one target grid per "
                                          f"new keyframe and rank, getFitnessScore(inf), record all-gather, best-candidate replay; inputs resident in HBM",
                              "parallelism": f"{world} x 1 GPU" if world > 1 else "1 GPU", "record_gather": backend if use_pg else None},
                   "roofline": r["roofline"], "roofline_fitness": r["roofline_fitness"], "cpu_baseline": None, "config3_shard": r}
            emit(out, args.full_line)
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- config[1] shape, weak scaling --------------------------------------------------------------------------------------
    scans, dev = to_hbm(raw)
    rels = [synth.rel_pose(poses[k], poses[k + 1]) for k in range(args.distinct)]
    pairs = []  # (target scan index, source scan index, guess, truth, cold)
    for b in range(args.batch):
        k = b % args.distinct
        cold = args.cold_every > 0 and b % args.cold_every == args.cold_every - 1
        guess = np.eye(4) if cold else synth.warm_guess(rels[k], 1000 * rank + b)
        pairs.append((k, k + 1, guess, rels[k], cold))
    raw_digest = sha16(raw)
    n_pts = float(np.mean([len(scans[p[1]]) for p in pairs]))
    n_src_per_step = float(sum(len(scans[p[1]]) for p in pairs)) / args.batch  # mean source points per alignment
    hbm_input_bytes = 16.0 * sum(len(scans[p[0]]) + len(scans[p[1]]) for p in pairs)
    t_gen = time.time() - t_gen

    per_mode = np.zeros((3, 3))  # [mode] -> (device ms, launches, algorithmic bytes), HIP events around every launch
    launched = np.zeros(2)       # (source points, valid point-voxel pairs) of the evaluations actually launched
    counters = {"evals": 0, "iters": 0, "on": False}
    points_by_kind = np.zeros(3)   # source points of the launched evaluations, per evaluation kind
    largest = {"ms": 0.0, "pairs": [0, 0, 0]}

    # one setInputTarget per alignment: every pair has its own target entry
    add_args = ([dev[p[0]].data_ptr() for p in pairs], [len(scans[p[0]]) for p in pairs], np.arange(args.batch, dtype=np.int32),
                [dev[p[1]].data_ptr() for p in pairs], [len(scans[p[1]]) for p in pairs], np.stack([p[2] for p in pairs]))

    def account(b, res):
        if use_pg:  # pose / Hessian record gather over RCCL
            res["pair_id"] = np.arange(args.batch, dtype=np.int32)
            all_gather_records(res, args.batch)
        if counters["on"]:
            for m in range(3):
                per_mode[m] += b.kernel_stats(m)
            launched[:] += np.array(b.pair_counts())
            for m in range(3):
                points_by_kind[m] += b.pair_counts(m)[0]
            lm, lp = b.largest_launch()
            if lm > largest["ms"]:
                largest["ms"], largest["pairs"] = lm, lp
            counters["evals"] += int(res["evaluations"].sum())
            counters["iters"] += int(res["iterations"].sum())
        return res

    def step():  # one step at a time: queue the batch, align it, take its records
        bm.clear()
        bm.add_device(*add_args)
        return account(bm, bm.align())

    # --in-flight N > 1 (default 2): N batches on N contexts, mrgfe_batch_align_async / mrgfe_batch_wait (a worker thread per batch behind the C
    # ABI).  A step then SUBMITS its batch and collects the records of the batch submitted N steps earlier: the target build and the straggler
    # rounds of one batch are filled by the derivative launches of the other.  Same pairs, same records.
    n_fl = max(1, args.in_flight)
    bms = [bm] + [BatchMatcher(prm, Context(local_rank)) for _ in range(n_fl - 1)]
    pending = [False] * n_fl
    slot = {"k": 0}

    def step_pipelined():
        k = slot["k"] % n_fl
        slot["k"] += 1
        b, res = bms[k], None
        if pending[k]:
            res = account(b, b.wait())
            pending[k] = False
        b.clear()
        b.add_device(*add_args)
        b.align_async()
        pending[k] = True
        return res

    def drain():
        last = None
        for j in range(n_fl):  # in submission order
            k = (slot["k"] + j) % n_fl
            if pending[k]:
                last = account(bms[k], bms[k].wait())
                pending[k] = False
        return last

    for _ in range(args.warmup):
        step()
    seq = None
    if n_fl > 1 and not args.no_seq:
        # the same K steps one at a time first (untimed for `value`; reported beside it): elapsed time, and the derivative kernel's HIP-event
        # time WITHOUT another batch's kernels on the chip — in the pipelined region every launch shares the chip with the other batch's
        counters["on"] = True
        e1, sm1, _ = timed(step, args.steps, 0)
        counters["on"] = False
        seq = {"value": world * args.batch * args.steps / e1, "unit": "alignments/s", "ms_per_step": 1e3 * e1 / args.steps, "steps": args.steps,
               "derivative_kernel_ms": float(per_mode[0][0]), "derivative_launches": int(per_mode[0][1]), "alg_bytes": float(per_mode[:, 2].sum()),
               "avg_launch_ms": float(per_mode[0][0] / per_mode[0][1]) if per_mode[0][1] else None,
               "frac": float((per_mode[:, 2].sum() / 1e9) / (per_mode[0][0] / 1e3) / HBM_PEAK_GBPS) if per_mode[0][0] > 0 else None,
               "note": "one step at a time on one context (the timed region of rounds 1-4): queue, align, records, then the next step"}
        per_mode[:] = 0
        launched[:] = 0
        points_by_kind[:] = 0
        largest["ms"], largest["pairs"] = 0.0, [0, 0, 0]
        counters["evals"] = counters["iters"] = 0
    if n_fl > 1:
        for _ in range(2):
            step_pipelined()
        drain()
    counters["on"] = True
    elapsed, step_ms, res = timed(step_pipelined if n_fl > 1 else step, args.steps, 0, drain if n_fl > 1 else None)
    counters["on"] = False
    evals, iters = counters["evals"], counters["iters"]
    print(f"[bench rank {rank}] per-step ms: " + " ".join(f"{v:.2f}" for v in step_ms), file=sys.stderr)
    total_pairs = world * args.batch * args.steps
    value = total_pairs / elapsed

    # the score + gradient + Hessian variant on its own (round 1 / 2 reported it as the dominant kernel): two extra, UNTIMED steps with one
    # launch per variant, so that its launches can be told from the others' — listed beside the fused kernel's figures, never `value`
    alone = None
    if lib().mrgfe_dbg_set_fused_launch(-1) == 1:
        lib().mrgfe_dbg_set_fused_launch(0)
        m0 = np.zeros(3)
        step()
        for _ in range(2):
            step()
            m0 += bm.kernel_stats(0)
        lib().mrgfe_dbg_set_fused_launch(1)
        if m0[1] and m0[0] > 0:
            alone = {"kernel": "ndt_derivatives_kernel<0,7>", "launches": int(m0[1]), "avg_launch_ms": m0[0] / m0[1], "alg_bytes_per_launch": m0[2] / m0[1],
                     "achieved_GBps": (m0[2] / 1e9) / (m0[0] / 1e3), "frac": (m0[2] / 1e9) / (m0[0] / 1e3) / HBM_PEAK_GBPS,
                     "note": "MRGFE_FUSED=0 (one launch per variant), 2 steps outside the timed region"}

    # ---- the rest of SURVEY.md §8(d), all outside the timed region above (never `value`) ------------------------------------------
    extras = {}
    if not args.no_extras and rank == 0 and world == 1:  # (single-GPU side measurements: not while the other ranks of an N-GPU run wait)
        # (1) setInputTarget and align apart (a synchronisation between them; the timed steps run them back to back)
        split = []
        for _ in range(3):
            bm.clear()
            bm.add_device(*add_args)
            ctx.synchronize()
            ta = time.perf_counter()
            bm.build_targets()
            ctx.synchronize()
            tb = time.perf_counter()
            bm.align()
            split.append((1e3 * (tb - ta), 1e3 * (time.perf_counter() - tb)))
        extras["gpu_split_ms_per_step"] = {"set_target_ms": float(np.median([a for a, _ in split])), "align_ms": float(np.median([b for _, b in split])),
                                           "pairs_per_step": args.batch, "note": "mrgfe_batch_build_targets then mrgfe_batch_align, device-resident clouds, median of 3 steps"}
        # (2) PCIe-inclusive rate: both clouds of every pair handed over as HOST pointers inside the timed region
        def host_step():
            bm.clear()
            for (ti, si, guess, _, _) in pairs:
                t = bm.add_target(scans[ti])
                bm.add_pair(t, scans[si], guess)
            return bm.align()

        host_step()
        ctx.synchronize()
        th = time.perf_counter()
        n_host = 3
        for _ in range(n_host):
            host_res = host_step()
        ctx.synchronize()
        th = time.perf_counter() - th
        extras["value_host_pointers"] = {"value": args.batch * n_host / th, "unit": "alignments/s", "ms_per_step": 1e3 * th / n_host, "steps": n_host,
                                         "same_results_as_device_pointers": bool(np.array_equal(host_res["T"], res["T"])),
                                         "note": f"mrgfe_batch_add_target + mrgfe_batch_add_pair with host clouds ({hbm_input_bytes / 1e6:.0f} MB over PCIe per step through the pinned staging ring) + align"}
        # (2b) the same with the host clouds page-locked (mrgfe_pin_host_buffer: a caller that keeps its keyframe clouds pinned): DMA straight out of them
        pinned = []
        try:
            for b in bms:
                b._ctx.set_zero_copy_uploads(True)
            for sc in scans:
                if lib().mrgfe_pin_host_buffer(ctx._h, sc.ctypes.data_as(C.c_void_p), sc.nbytes) == 0:
                    pinned.append(sc)
            if len(pinned) == len(scans):
                host_step()
                ctx.synchronize()
                tp = time.perf_counter()
                for _ in range(n_host):
                    pin_res = host_step()
                ctx.synchronize()
                tp = time.perf_counter() - tp
                # ... and two batches in flight (mrgfe_batch_align_async): the uploads of one batch go over the link while the other aligns
                pin_pipe = None
                if n_fl > 1:
                    def host_submit(b):
                        b.clear()
                        for (ti, si, guess, _, _) in pairs:
                            b.add_pair(b.add_target(scans[ti]), scans[si], guess)
                        b.align_async()

                    n_pipe = 6
                    host_submit(bms[0])
                    bms[0].wait()
                    ctx.synchronize()
                    tq = time.perf_counter()
                    live = [False] * n_fl
                    pipe_res = None
                    for it in range(n_pipe):
                        k = it % n_fl
                        if live[k]:
                            pipe_res = bms[k].wait()
                        host_submit(bms[k])
                        live[k] = True
                    for j in range(n_fl):
                        k = (n_pipe + j) % n_fl
                        if live[k]:
                            pipe_res = bms[k].wait()
                            live[k] = False
                    tq = time.perf_counter() - tq
                    pin_pipe = {"value": args.batch * n_pipe / tq, "unit": "alignments/s", "ms_per_step": 1e3 * tq / n_pipe, "steps": n_pipe, "steps_in_flight": n_fl,
                                "GBps_over_pcie": hbm_input_bytes / 1e9 / (tq / n_pipe), "same_results_as_device_pointers": bool(np.array_equal(pipe_res["T"], res["T"]))}
                extras["value_host_pointers"]["pinned_host_clouds"] = {"two_batches_in_flight": pin_pipe, "value": args.batch * n_host / tp, "unit": "alignments/s", "ms_per_step": 1e3 * tp / n_host,
                                                                       "GBps_over_pcie": hbm_input_bytes / 1e9 / (tp / n_host), "same_results_as_device_pointers": bool(np.array_equal(pin_res["T"], res["T"])),
                                                                       "note": "the caller's clouds page-locked once (mrgfe_pin_host_buffer): uploads are DMA out of the caller's buffers, no staging copy"}
        finally:
            for b in bms:
                b._ctx.set_zero_copy_uploads(False)
            for sc in pinned:
                lib().mrgfe_unpin_host_buffer(ctx._h, sc.ctypes.data_as(C.c_void_p))
        # (3) pipeline shape: what prefiltering_component -> scan_matching_odometry really feeds NDT (distance + 0.1 m voxel + radius outlier filter)
        if args.prefilter != "full" and raw is not None:
            f_host = [prefilter(sc, ctx=ctx) for sc in raw]
            f_dev = [torch.from_numpy(sc).to(gdev) for sc in f_host]
            f_args = ([f_dev[p[0]].data_ptr() for p in pairs], [len(f_host[p[0]]) for p in pairs], np.arange(args.batch, dtype=np.int32),
                      [f_dev[p[1]].data_ptr() for p in pairs], [len(f_host[p[1]]) for p in pairs], np.stack([p[2] for p in pairs]))

            def f_step():
                bm.clear()
                bm.add_device(*f_args)
                return bm.align()

            f_step()
            ctx.synchronize()
            tf = time.perf_counter()
            n_f = 5
            for _ in range(n_f):
                f_res = f_step()
            ctx.synchronize()
            tf = time.perf_counter() - tf
            f_err = [float(np.linalg.norm(result_matrix(f_res[b])[:3, 3] - pairs[b][3][:3, 3])) for b in range(args.batch)]
            extras["pipeline_shape"] = {"value": args.batch * n_f / tf, "unit": "alignments/s", "ms_per_step": 1e3 * tf / n_f, "steps": n_f,
                                        "mean_points_per_scan": float(np.mean([len(c) for c in f_host])), "iterations_per_alignment": float(f_res["iterations"].mean()),
                                        "median_translation_error_vs_truth_m": float(np.median(f_err)),
                                        "note": "the same pairs after mrgfe_prefilter (distance 0.1-35 m, VoxelGrid 0.1 m, RadiusOutlierRemoval 0.5 m / 2): the synthetic street saturates "
                                                "that voxel grid at a quarter of the points"}
            del f_dev
        # (3b) what the reference's own summation order costs (opt-in, mrgfe_dbg_set_ndt_reference_order): the first 64 pairs of the step, host-stepped, record +
        # chain kernels, against the same 64 pairs in the default order
        n_ro = min(64, args.batch)
        ro_args = tuple(a[:n_ro] for a in add_args)

        def ro_step():
            bm.clear()
            bm.add_device(*ro_args)
            return bm.align()

        ro_t = {}
        for mode in (0, 1):
            lib().mrgfe_dbg_set_ndt_reference_order(mode)
            try:
                ro = ro_step()
                ctx.synchronize()
                tr = time.perf_counter()
                for _ in range(2):
                    ro = ro_step()
                ctx.synchronize()
                ro_t[mode] = ((time.perf_counter() - tr) / 2, ro)
            finally:
                lib().mrgfe_dbg_set_ndt_reference_order(0)
        extras["ndt_reference_order"] = {"pairs": n_ro, "ms_per_step": 1e3 * ro_t[1][0], "ms_per_step_default_order": 1e3 * ro_t[0][0], "slowdown": ro_t[1][0] / ro_t[0][0],
                                         "pairs_with_the_default_orders_transformation": int((ro_t[1][1]["T"] == ro_t[0][1]["T"]).all(axis=1).sum()),
                                         "same_iterations_and_convergence_as_default": bool(np.array_equal(ro_t[1][1]["iterations"], ro_t[0][1]["iterations"]) and
                                                                                            np.array_equal(ro_t[1][1]["converged"], ro_t[0][1]["converged"])),
                                         "note": "MRGFE_NDT_REFERENCE_ORDER=1: per-point sums + point-order chains (computeDerivatives), pair-order chain (computeHessian), Eigen's "
                                                 "JacobiSVD solve; bit-identical to the reference-order oracle (tests/test_gpu_ndt_reforder.py, soak_over_bar.ndt_reference_order); the "
                                                 "chains are one wavefront per evaluation whatever the batch size (~1 ms per 130k-point evaluation, ~9 ms per f64 Hessian pass)"}
        # (4) BASELINE config[2]: GICP scan-to-keyframe (the k-NN correspondence path), keyframe = scan 0, frames = scans 1..6
        extras["config2_gicp"] = run_config2(ctx, scans, dev, poses, lib, args)
        # (4b) registration_method "NDT": pcl::NormalDistributionsTransform, the f64 formulation, on the headline's pairs
        extras["pcl_ndt"] = run_pcl_ndt(ctx, scans, dev, pairs, lib, args)
        # (5) the per-scan path every robot of config[4] runs: raw scan in host memory -> mrgfe_prefilter_device (distance 0.1-35 m, VoxelGrid 0.1 m,
        # RadiusOutlierRemoval 0.5 m / 2) -> setInputSourceDevice + scan-to-keyframe align, with the chain's counts on the device (one host wait) and, for
        # comparison, every stage reporting to the host (round 3)
        if raw is not None:
            from mrg_slam_amd import prefilter_to_device

            dbuf = torch.empty((max(len(r) for r in raw[:8]) + 16, 4), dtype=torch.float32, device=gdev)
            kf = prefilter(raw[0], ctx=ctx)
            odo = NdtHip(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, ctx=ctx)
            odo.setInputTarget(kf)
            per_scan = {}
            for mode, name in ((1, "device_driven"), (0, "host_driven")):
                lib().mrgfe_dbg_set_prefilter_device_driven(mode)
                t_pf, t_fr = [], []
                for rep in range(3):
                    for k in range(1, 7):
                        g = synth.warm_guess(synth.rel_pose(poses[0], poses[k]), 9000 + k)
                        ctx.synchronize()
                        t0 = time.perf_counter()
                        m = prefilter_to_device(raw[k], dbuf.data_ptr(), dbuf.shape[0], ctx=ctx)
                        t1 = time.perf_counter()
                        odo.setInputSourceDevice(dbuf.data_ptr(), m)
                        odo.align(g)
                        t2 = time.perf_counter()
                        if rep:
                            t_pf.append(1e3 * (t1 - t0))
                            t_fr.append(1e3 * (t2 - t0))
                per_scan[name] = {"prefilter_ms": float(np.median(t_pf)), "prefilter_plus_ndt_frame_ms": float(np.median(t_fr))}
            lib().mrgfe_dbg_set_prefilter_device_driven(1)
            # the same path with the reference's YAML default, registration_method "SMALL_GICP" (config/mrg_slam.yaml:100), as the odometry runs it on this
            # 1 m / scan trajectory: every frame is aligned against the frame before it and then becomes the keyframe (keyframe_delta_translation 1.0, :80;
            # scan_matching_odometry_component.cpp:326-339) — taken over from the source (mrgfe_reg_source_becomes_target)
            from mrg_slam_amd import SmallGicpHip

            gic = SmallGicpHip(transformation_epsilon=args.eps, ctx=ctx)
            bufs = [torch.empty_like(dbuf), torch.empty_like(dbuf)]
            t_g = []
            for rep in range(3):
                m0 = prefilter_to_device(raw[0], bufs[0].data_ptr(), bufs[0].shape[0], ctx=ctx)
                gic.setInputTargetDevice(bufs[0].data_ptr(), m0)
                for k in range(1, 7):
                    g = synth.warm_guess(synth.rel_pose(poses[k - 1], poses[k]), 9500 + k)
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    b = bufs[k % 2]
                    mk = prefilter_to_device(raw[k], b.data_ptr(), b.shape[0], ctx=ctx)
                    gic.setInputSourceFromPrefilter(b.data_ptr(), mk)
                    gic.align(g)
                    gic.sourceBecomesTarget()
                    if rep:
                        t_g.append(1e3 * (time.perf_counter() - t0))
            per_scan["device_driven"]["prefilter_plus_small_gicp_frame_ms"] = float(np.median(t_g))
            per_scan["raw_points"] = int(np.mean([len(r) for r in raw[1:7]]))
            per_scan["filtered_points"] = int(m)
            per_scan["note"] = ("median of 12 frames, raw scans handed over as host pointers (the 2 MB upload is inside), result left in HBM for the scan matcher; NDT_HIP frames "
                                "against one keyframe, SMALL_GICP_HIP frames each against the frame before it, which it then takes over as its keyframe")
            extras["per_scan_path"] = per_scan

    shard = None
    if args.shard_steps > 0:
        shard = run_shard(args.shard_steps, 1)
        if not args.no_extras and rank == 0 and world == 1:
            shard["detect_batched"] = run_detect_leg(ctx, prm, loop_raw)

    if rank != 0:
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- single-pair latency (one pcl::Registration-style object), opt-in ------------------------------------------------------
    single_ms = None
    if not args.no_latency:
        reg = NdtHip(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, ctx=ctx)
        ti, si, guess, truth, _ = pairs[0]
        lat = []
        for _ in range(12):
            ctx.synchronize()
            t1 = time.perf_counter()
            reg.setInputTargetDevice(dev[ti].data_ptr(), len(scans[ti]))
            reg.setInputSourceDevice(dev[si].data_ptr(), len(scans[si]))
            reg.align(guess)
            lat.append(time.perf_counter() - t1)
        single_ms = 1e3 * float(np.median(lat[2:]))
    # mean valid neighbour voxels per point of the evaluations actually launched (k-bar of SURVEY.md §8d); evaluations a
    # controller answers from its cache (repeated line-search trials) are neither launched nor counted
    kbar = float(launched[1] / launched[0]) if launched[0] else 0.0
    # ---- CPU baseline + parity on a bounded sample of the same pairs ---------------------------------------------------
    cpu = None
    parity = None
    if not args.no_cpu and args.cpu_pairs > 0 and rank == 0 and world == 1:  # the CPU baseline and the parity check: rank 0 of the 1-GPU run only
        from oracle import oracle as orc

        host_cores = os.cpu_count() or 1
        ncpu = min(args.cpu_pairs, len(pairs))
        # thread count: the reference default (reg_num_threads: 8, config/mrg_slam.yaml:101) and wider settings up to the
        # host's cores; the fastest one is reported (the per-point OpenMP loop stops scaling well before 256 threads), the
        # reference default beside it
        sweep = sorted({t for t in (8, 16, 32, 64, host_cores) if t <= host_cores})

        def run_cpu(nt, sample, thread_sums=True):
            # thread_sums: ndt_omp's own accumulation (one accumulator per OpenMP thread) — what is TIMED; the checker's per-point records
            # added in point order (thread-count invariant, but 45 MB written and summed serially per evaluation) would not scale at all
            o = orc.Ndt(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, num_threads=nt, thread_sums=thread_sums)
            t_set = t_align = 0.0
            out = []
            for (ti, si, guess, _, _) in sample:
                t0 = time.perf_counter()
                o.setInputTarget(scans[ti])
                t1 = time.perf_counter()
                o.setInputSource(scans[si])
                o.align(guess)
                t2 = time.perf_counter()
                t_set += t1 - t0
                t_align += t2 - t1
                out.append((o.getFinalTransformation(), o.hasConverged(), o.getFinalNumIteration(), o.evals))
            return t_set + t_align, t_set, t_align, out

        probe = pairs[:min(8, ncpu)]  # thread-count probe on a few pairs, then the whole sample with the fastest setting
        probe_r = {nt: run_cpu(nt, probe) for nt in sweep}
        cores = min(sweep, key=lambda nt: probe_r[nt][0])
        tc, tc_set, tc_align, o_res = run_cpu(cores, pairs[:ncpu])
        cpu = {"value": ncpu / tc, "unit": "alignments/s", "cores": cores, "kind": "port",
               "sample_short": f"{ncpu} of the step's {args.batch} pairs (setInputTarget + align each), restated pclomp NDT_OMP, {cores} OpenMP threads (fastest of {sweep}), {tc:.1f} s",
               "sample": f"{ncpu} of the {args.batch} pairs of one step (setInputTarget+align each, same warm/cold guesses), CPU oracle = restated pclomp NDT_OMP "
                         f"(not the upstream library: parity unpinned, DESIGN.md §2), -O3 -fopenmp, fastest of OpenMP thread counts {sweep} on a {host_cores}-thread "
                         f"host = {cores} threads, {tc:.2f} s",
               "set_target_ms_per_pair": 1e3 * tc_set / ncpu, "align_ms_per_pair": 1e3 * tc_align / ncpu,
               "reference_default_8_threads": {"value": len(probe) / probe_r[8][0], "pairs": len(probe), "set_target_ms_per_pair": 1e3 * probe_r[8][1] / len(probe),
                                               "align_ms_per_pair": 1e3 * probe_r[8][2] / len(probe), "note": "reg_num_threads: 8 (config/mrg_slam.yaml:101)"} if 8 in probe_r else None,
               "split_note": "timed with ndt_omp's accumulation (one score / gradient / Hessian accumulator per OpenMP thread, static chunks); setInputTarget "
                             "(VoxelGridCovariance::filter) and computeHessian are serial in ndt_omp and in the restatement",
               "host_cpu": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "unknown")}
        # parity on ALL pairs of the step (the timing above is the sample; the check is not): the remaining pairs at the same thread count, untimed
        n_par = len(pairs) if args.parity_pairs <= 0 else min(args.parity_pairs, len(pairs))
        if n_par > ncpu:
            o_res = o_res + run_cpu(cores, pairs[ncpu:n_par], thread_sums=False)[3]
        n_par = len(o_res)
        dts, drs, over, mism, capped = [], [], 0, 0, 0
        for k in range(n_par):
            Tg = result_matrix(res[k])
            To, conv, it, ev = o_res[k]
            dts.append(float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])))
            drs.append(rot_angle(Tg[:3, :3], To[:3, :3]))
            over += int(dts[-1] > 1e-4 or drs[-1] > 1e-4)
            mism += int(bool(res[k]["converged"]) != conv or int(res[k]["iterations"]) != it)
            capped += int(it > 64)
        parity = {"pairs": n_par, "max_dt_m": max(dts), "max_dr_rad": max(drs), "pairs_over_bar": over, "pairs_with_other_iterations_or_convergence": mism,
                  "pairs_at_the_iteration_limit": capped, "same_iterations_and_convergence": mism == 0, "bar": "1e-4 m / 1e-4 rad"}

    # ---- every tolerated over-the-bar case as a number: the randomised soak of the GPU suite, a bounded sample of it per run
    soak = None
    if not args.no_cpu and args.soak_cases > 0 and rank == 0 and world == 1:
        from oracle.replay import ndt_soak, pclndt_soak, round3_soak

        ts = time.perf_counter()
        a = ndt_soak(args.soak_cases, 20260411)
        b = round3_soak(max(1, args.soak_cases // 3), 20260412)
        c = pclndt_soak(max(1, args.soak_cases // 2), 20260413)
        # ... and NDT_HIP with its sums and Newton solve in the reference's order (opt-in mode, mrgfe_dbg_set_ndt_reference_order): bit identity is the bar there
        from oracle.replay import ndt_reference_order_soak

        lib().mrgfe_dbg_set_ndt_reference_order(1)
        try:
            tr = time.perf_counter()
            d = ndt_reference_order_soak(max(1, args.soak_cases // 2), 20260414)
            d["seconds"] = time.perf_counter() - tr
        finally:
            lib().mrgfe_dbg_set_ndt_reference_order(0)
        soak = {"ndt": f"{a['ndt_over_bar']}/{a['ndt']}", "ndt_settled": f"{a['ndt_settled_over_bar']}/{a['ndt_settled']}",
                "ndt_over_bar_equal_to_gpu_order_replay": f"{a['ndt_over_bar_equal_to_gpu_order_replay']}/{a['ndt_over_bar']}",
                "ndt_bit_identical_to_reference_order_oracle": f"{a['ndt_exact_ref']}/{a['ndt']}", "ndt_bit_identical_to_gpu_order_replay": f"{a['ndt_exact_gpu_order']}/{a['ndt']}",
                "ndt_worst_settled_m_or_rad": a["ndt_worst_settled"], "ndt_worst_m_or_rad": a["ndt_worst"],
                "icp_gicp_vgicp_small_gicp": f"{a['other_over_bar']}/{a['other']}", "icp_gicp_vgicp_small_gicp_bit_identical": f"{a['other_exact']}/{a['other']}",
                "pcl_gicp_serial": f"{b['gicp_serial_over_bar']}/{b['gicp_serial']}", "pcl_gicp_serial_bit_identical_to_reference_order_oracle": f"{b['gicp_serial_exact_ref']}/{b['gicp_serial']}",
                "pcl_gicp_serial_worst_m_or_rad": b["gicp_serial_worst"],
                "pcl_gicp_omp": f"{b['gicp_omp_over_bar']}/{b['gicp_omp']}", "pcl_gicp_omp_over_bar_equal_to_gpu_order_replay": f"{b['gicp_omp_over_bar_equal_to_gpu_order_replay']}/{b['gicp_omp_over_bar']}",
                "pcl_gicp_omp_bit_identical_to_gpu_order_replay": f"{b['gicp_omp_exact_gpu_order']}/{b['gicp_omp']}", "pcl_gicp_omp_bit_identical_to_reference_order_oracle": f"{b['gicp_omp_exact_ref']}/{b['gicp_omp']}",
                "pcl_gicp_omp_worst_m_or_rad": b["gicp_omp_worst"], "icp_reciprocal": f"{b['icp_over_bar']}/{b['icp']}",
                "ndt_reference_order": f"{d['over_bar']}/{d['cases']}", "ndt_reference_order_bit_identical": f"{d['exact']}/{d['cases']}",
                "ndt_reference_order_unsettled_scenes": d["unsettled"], "ndt_reference_order_seconds": d["seconds"],
                "pcl_ndt": f"{c['over_bar']}/{c['cases']}", "pcl_ndt_bit_identical_to_reference_order_oracle": f"{c['exact']}/{c['cases']}", "pcl_ndt_worst_m_or_rad": c["worst"],
                "pcl_ndt_scenes_that_stop_after_one_iteration": f"{c['one_iteration']}/{c['cases']}",
                "worst_m": max(a["ndt_worst"], a["other_worst"], b["gicp_serial_worst"], b["gicp_omp_worst"], b["icp_worst"], c["worst"]),
                "flag_or_iteration_mismatches": a["ndt_flag_or_iteration_mismatch"] + a["other_flag_mismatch"] + b["gicp_flag_or_iteration_mismatch"] + b["icp_flag_or_iteration_mismatch"]
                                                + c["flag_or_iteration_mismatch"],
                "over_bar_cases": a["over_bar"] + b["over_bar"] + c["over_bar_cases"], "seconds": time.perf_counter() - ts,
                "what": f"{args.soak_cases} + {max(1, args.soak_cases // 3)} random 1.5k-9k-point scenes (oracle/replay.py, seeds 20260411 / 20260412: every method, resolution 0.5-2 m, four NDT "
                        "neighbourhoods, eps 0.1-0.001, warm and identity guesses), k/N = scenes over the 1e-4 m / 1e-4 rad bar against the reference-order oracle; "
                        "'settled' = converged within 30 iterations on both sides; a larger run of the same soak is kept under profiles/"}

    # dominant kernel.  Fused launches (default): ndt_derivatives_all_kernel<7>, ONE launch per round that runs the work items of all
    # three evaluation kinds (score + gradient + Hessian, score + gradient, f64 Hessian): its time and launch count are reported
    # under variant 0 by the library, its algorithmic bytes are those of all three kinds.  MRGFE_FUSED=0: ndt_derivatives_kernel<0,7>
    # (score + gradient + Hessian), the other two variants listed beside it.
    fused = bool(lib().mrgfe_dbg_set_fused_launch(-1))
    k_ms, k_launch, k_bytes = per_mode[0]
    if fused:
        k_bytes = float(per_mode[:, 2].sum())
    kernel_name = "ndt_derivatives_all_kernel<7>" if fused else "ndt_derivatives_kernel<0,7>"
    achieved = (k_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
    variants = {name: {"launches": int(per_mode[m][1]), "avg_launch_ms": (per_mode[m][0] / per_mode[m][1]) if per_mode[m][1] else None,
                       "achieved_GBps": (per_mode[m][2] / 1e9) / (per_mode[m][0] / 1e3) if per_mode[m][0] > 0 else None}
                for m, name in enumerate(("ndt_derivatives_kernel<0,7>", "ndt_derivatives_kernel<1,7>", "ndt_derivatives_kernel<2,7>"))
                if per_mode[m][1] and not fused}  # (the line-search and f64-Hessian variants are timed only with MRGFE_KERNEL_TIMING=2: their events cost the round a little)
    # HBM bytes per launch of the dominant kernel and its VALU utilisation from the PMC passes (collected separately with
    # rocprofv3 --pmc and corrected as MI355X_MICROARCH.md prescribes; profiles/summarize.py) - null until a profile exists
    traffic = valu_busy = None
    prof_name = None
    prof_name, pj = latest_pmc_summary()
    traffic = pj.get("traffic_bytes_per_mean_launch")
    valu_busy = pj.get("valu_busy_dominant")
    alg_per_launch = (k_bytes / k_launch) if k_launch else None
    # the largest launch of the timed steps (a trace shows it as the kernel's maximum): its round's busy pairs per kind x the mean algorithmic
    # bytes of one evaluation of that kind
    largest_rec = None
    if largest["ms"] > 0 and n_src_per_step > 0:
        per_eval = [per_mode[m][2] / (points_by_kind[m] / n_src_per_step) if points_by_kind[m] > 0 else 0.0 for m in range(3)]
        lb = float(sum(n * b for n, b in zip(largest["pairs"], per_eval)))
        largest_rec = {"ms": largest["ms"], "busy_pairs_by_kind": largest["pairs"], "alg_bytes": lb, "achieved_GBps": lb / 1e9 / (largest["ms"] / 1e3),
                       "frac": lb / 1e9 / (largest["ms"] / 1e3) / HBM_PEAK_GBPS}
    ratio = (traffic / alg_per_launch) if (traffic and alg_per_launch) else None
    # what the counters say limits the kernel: HBM only if the bytes it really moves are a large share of the byte model
    limiter = "hbm" if (ratio is None or ratio >= 0.5) else "valu"
    true_err = [float(np.linalg.norm(result_matrix(res[b])[:3, 3] - pairs[b][3][:3, 3])) for b in range(args.batch)]
    cold_ix = [b for b in range(args.batch) if pairs[b][4]]
    warm_ix = [b for b in range(args.batch) if not pairs[b][4]]
    out = {
        "metric": METRIC,
        "value": value,
        "unit": "alignments/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 per-pair terms, f64 accumulation",
        "data": "synthetic" if scene is not None else "KITTI odometry sequence 00 (KITTI_ROOT)",
        "config": {
            "workload": f"BASELINE config[1] shape: synthetic VLP-64 scan-to-scan NDT_HIP (DIRECT7, resolution 1.0 m, eps {args.eps}, max_iter 64), "
                        f"{args.batch} pairs per GPU per step = {min(args.distinct, args.batch)} distinct scan pairs ({hbm_input_bytes / 1e6:.0f} MB of clouds per step), "
                        f"{len(warm_ix)} warm + {len(cold_ix)} cold (identity) guesses, mean {n_pts:.0f} pts/scan "
                        f"({'distance filter 0.1-35 m' if args.prefilter == 'distance' else 'distance + 0.1 m voxel + radius outlier prefilter'}), "
                        f"setInputTarget + setInputSource + align per pair, inputs resident in HBM"
                        + (f"; {n_fl} steps in flight per GPU (mrgfe_batch_align_async on {n_fl} contexts: a step submits its batch and collects the one submitted "
                           f"{n_fl} steps earlier; every submitted step completes inside the timed region)" if n_fl > 1 else ""),
            "workload_short": f"BASELINE config[1]: {args.batch} distinct synthetic VLP-64 scan pairs per GPU per step, NDT_HIP DIRECT7 res 1.0 eps {args.eps} max_iter 64, "
                              f"setInputTarget + setInputSource + align per pair, clouds resident in HBM, {n_fl} step(s) in flight",
            "steps_in_flight": n_fl,
            "pairs_per_gpu_per_step": args.batch,
            "distinct_pairs": min(args.distinct, args.batch),
            "points_per_scan": n_pts,
            "parallelism": f"{world} x 1 GPU, own pairs per rank, RCCL all-gather of 384-byte result records" if world > 1 else "1 GPU",
            "record_gather": backend if use_pg else None,
        },
        "roofline": {"bound": limiter, "byte_model_bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_over_algorithmic": ratio, "valu_busy": valu_busy, "pmc_profile": prof_name,
                     "kernel": kernel_name, "avg_launch_ms": (k_ms / k_launch) if k_launch else None, "launches": int(k_launch),
                     "alg_bytes_per_launch": alg_per_launch,
                     "byte_model": "per launch: sum over the evaluations of all active pairs (every kind: score+gradient+Hessian, score+gradient, f64 Hessian) of "
                                   "N_src*(16 + 7*8) + valid_neighbours*48 (SURVEY.md §8d); `frac` prices these ALGORITHMIC bytes against the HBM peak as the contract "
                                   "asks; `bound` is what the PMC counters say limits the kernel (most of these bytes are served by L2 / Infinity Cache: `traffic`)",
                     "alg_bytes_by_evaluation_kind": {"score+gradient+hessian": per_mode[0][2] / max(k_launch, 1), "score+gradient": per_mode[1][2] / max(k_launch, 1),
                                                      "f64_hessian": per_mode[2][2] / max(k_launch, 1)},
                     "launches_overlap": n_fl > 1,
                     "aggregate_GBps_over_the_timed_region": (k_bytes / 1e9) / elapsed if elapsed > 0 else None,
                     "aggregate_frac_over_the_timed_region": (k_bytes / 1e9) / elapsed / HBM_PEAK_GBPS if elapsed > 0 else None,
                     "one_step_at_a_time": ({"avg_launch_ms": seq["avg_launch_ms"], "launches": seq["derivative_launches"], "frac": seq["frac"],
                                             "achieved": seq["frac"] * HBM_PEAK_GBPS if seq["frac"] else None} if seq else None),
                     "overlap_note": (f"{n_fl} batches are in flight in the timed region: a derivative launch shares the chip with the other batch's launches, so its HIP-event "
                                      "duration (and `frac`, which prices ONE launch's algorithmic bytes against it) is about that of two launches side by side, and the sum of "
                                      "the launch durations exceeds the wall time of the region; `aggregate_*` = algorithmic bytes of ALL derivative launches of the region over "
                                      "its wall time (which also holds the target builds and the controller kernels); `one_step_at_a_time` = the same kernel in the K steps "
                                      "run before the timed region on one context with nothing beside it (rounds 1-4's region)") if n_fl > 1 else None,
                     "largest_launch": largest_rec,
                     "score_gradient_hessian_variant_alone": alone,
                     "variants": variants},
        "pmc_reference": pmc_reference(prof_name, pj),
        "cpu_baseline": cpu,
        "parity_vs_oracle": parity,
        "soak_over_bar": soak,
        "raw_inputs_sha256_16": raw_digest,
        "raw_inputs_as_in_the_build_container": (raw_digest == EXPECTED_RAW_INPUTS["config1_rank0_257_scans"]) if (EXPECTED_RAW_INPUTS["config1_rank0_257_scans"] and rank == 0 and args.distinct == 256 and scene is not None) else None,
        "evaluations_per_alignment": evals / (args.batch * args.steps),
        "evaluations_launched_per_alignment": float(launched[0] / (n_src_per_step * args.batch * args.steps)) if n_src_per_step else None,
        "iterations_per_alignment": iters / (args.batch * args.steps),
        "iterations_warm_cold": [float(np.mean(res["iterations"][warm_ix])) if warm_ix else None, float(np.mean(res["iterations"][cold_ix])) if cold_ix else None],
        "converged": int(res["converged"].sum()),
        "mean_valid_neighbours": kbar,
        "single_pair_latency_ms": single_ms,
        "median_translation_error_vs_truth_m": float(np.median(true_err)),
        "input_generation_s": t_gen,
        "steps_in_flight": n_fl,
        "value_one_step_at_a_time": seq,
        "value_host_pointers": extras.get("value_host_pointers"),
        "gpu_split_ms_per_step": extras.get("gpu_split_ms_per_step"),
        "pipeline_shape": extras.get("pipeline_shape"),
        "ndt_reference_order": extras.get("ndt_reference_order"),
        "config2_gicp": extras.get("config2_gicp"),
        "pcl_ndt": extras.get("pcl_ndt"),
        "per_scan_path": extras.get("per_scan_path"),
        "config3_shard": shard,
    }
    emit(out, args.full_line)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
