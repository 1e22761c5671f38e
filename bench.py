#!/usr/bin/env python3
"""bench.py — scan-pair NDT alignments per second on MI355X (BASELINE.json metric), one rank per GPU.

Step      = one pass of the hot path over one batch of B (default 256: the candidate-batch size of BASELINE config[3])
            synthetic VLP-64 scan pairs already resident in HBM:
            for every pair  setInputTarget (voxel covariance grid build)  +  setInputSource  +  align(guess)
            (reference call sites: apps/scan_matching_odometry_component.cpp:203,208,265-266; loop_detector.cpp:104,127,134),
            advanced together by the batched engine (one derivative launch per round for all pairs still running).
Workload  = BASELINE config[1] shape: ~120k points per scan, NDT resolution 1.0 m, DIRECT7, reg_transformation_epsilon 0.1,
            reg_maximum_iterations 64 (config/mrg_slam.yaml:100-109), warm initial guesses (perturbed truth, seed 777+k).
            No KITTI data exists here: scans are ray-cast by mrg_slam_amd/synth.py; the 0.1 m voxel
            prefilter saturates this synthetic street at ~35k points, so the ~120k-point clouds the metric is quoted
            on are the distance-filtered (0.1..35 m) scans (--prefilter full selects the whole chain instead).
N > 1     = weak scaling: every rank aligns its own B pairs (independent units, no data-path collective), then the ranks
            all-gather the 384-byte result records over RCCL (the pose/Hessian gather of the north star).
Output    = ONE JSON line (rank 0) with `roofline` (derivative kernel: algorithmic bytes / HIP-event time on the launch
            stream) and `cpu_baseline` (the CPU oracle, kind "port", timed on a bounded sample of the same pairs).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def rot_angle(Ra, Rb):
    from mrg_slam_amd import synth

    return synth.rotation_angle(Ra, Rb)


def make_workload(n_distinct: int, batch: int, rank: int, prefilter_mode: str):
    """Returns (targets, sources, guesses, truths): `batch` pairs built from `n_distinct` consecutive synthetic scans."""
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
    scene = synth.street_scene()
    # every rank drives its own stretch of the street (40 m further along x), same gentle arc
    start = synth.make_pose([40.0 * rank, 0.0, 0.0], np.eye(3))
    poses = [start @ T for T in synth.arc_trajectory(n_distinct + 1)]
    scans = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + 40 * rank + k) for k in range(n_distinct + 1)]
    return scene, poses, scans


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="scan pairs per step and per GPU (more pairs in flight keep the GPU full in the late rounds)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic scan pairs generated per rank (reused with different guesses)")
    ap.add_argument("--prefilter", choices=["distance", "full"], default="distance")
    ap.add_argument("--eps", type=float, default=0.1, help="reg_transformation_epsilon (config/mrg_slam.yaml:102)")
    ap.add_argument("--cpu-pairs", type=int, default=96, help="pairs of the bounded CPU-oracle sample, ~10-15 s of CPU work (0 disables)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--latency", action="store_true", help="also time single-pair setInputTarget+align latency (extra, differently sized launches of the "
                                                            "same kernels: off by default so rocprof averages of the default run describe the timed workload)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch  # first: libmrgfe then binds to the HIP runtime already in the process
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("BENCH_DIST_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()  # test hook only: gloo ranks may share a GPU (one rank per GPU otherwise)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; BENCH_DIST_BACKEND=gloo lets two ranks share one GPU to exercise this path on a 1-GPU box (RCCL
        # refuses duplicate devices)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from mrg_slam_amd import BatchMatcher, Context, NdtHip, distance_filter, prefilter, synth
    from mrg_slam_amd._lib import NDT_HIP, SEARCH
    from mrg_slam_amd.registration import RESULT_DTYPE, default_params, result_matrix

    ctx = Context(local_rank)

    # ---- inputs (untimed): synthetic scans -> prefilter on the GPU -> resident in HBM -------------------------------
    t_gen = time.time()
    scene, poses, raw = make_workload(args.distinct, args.batch, rank, args.prefilter)
    scans = [prefilter(s, ctx=ctx) if args.prefilter == "full" else distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
    rels = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(args.distinct)]
    dev = [torch.from_numpy(s).cuda(local_rank) for s in scans]
    pairs = []  # (target scan index, source scan index, guess, truth)
    for b in range(args.batch):
        k = b % args.distinct
        guess = synth.warm_guess(rels[k], 1000 * rank + b)
        pairs.append((k, k + 1, guess, rels[k]))
    n_pts = float(np.mean([len(scans[p[1]]) for p in pairs]))
    t_gen = time.time() - t_gen

    prm = default_params(NDT_HIP)
    prm.transformation_epsilon = args.eps
    prm.maximum_iterations = 64
    prm.resolution = 1.0
    prm.nn_search_method = SEARCH["DIRECT7"]
    prm.num_threads = 8
    bm = BatchMatcher(prm, ctx)
    gathered = torch.empty((world * args.batch, RESULT_DTYPE.itemsize), dtype=torch.uint8, device=f"cuda:{local_rank}") if world > 1 else None

    def step():
        bm.clear()
        for (ti, si, guess, _) in pairs:
            t = bm.add_target_device(dev[ti].data_ptr(), len(scans[ti]))  # one setInputTarget per alignment
            bm.add_pair_device(t, dev[si].data_ptr(), len(scans[si]), guess)
        res = bm.align()
        if world > 1:  # pose / Hessian record gather over RCCL
            mine = torch.from_numpy(res.view(np.uint8).reshape(args.batch, -1)).cuda(local_rank)
            dist.all_gather_into_tensor(gathered, mine)
        return res

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    # a full (generation 2) collection walks every object `import torch` created: ~40 ms, once, at an arbitrary step.
    # Collect now and move the survivors to the permanent generation so the timed steps are not interrupted by it.
    gc.collect()
    gc.freeze()
    per_mode = np.zeros((3, 3))  # [mode] -> (device ms, launches, algorithmic bytes), HIP events around every launch
    launched = np.zeros(2)       # (source points, valid point-voxel pairs) of the evaluations actually launched
    evals = iters = 0
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        res = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
        for m in range(3):
            per_mode[m] += bm.kernel_stats(m)
        launched += np.array(bm.pair_counts())
        evals += int(res["evaluations"].sum())
        iters += int(res["iterations"].sum())
    sync()
    elapsed = time.perf_counter() - t0
    print(f"[bench rank {rank}] per-step ms: " + " ".join(f"{v:.2f}" for v in step_ms), file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_pairs = world * args.batch * args.steps
    value = total_pairs / elapsed

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- single-pair latency (one pcl::Registration-style object, host loop per evaluation), opt-in ------------------
    single_ms = None
    if args.latency:
        reg = NdtHip(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, ctx=ctx)
        ti, si, guess, truth = pairs[0]
        lat = []
        for _ in range(6):
            ctx.synchronize()
            t1 = time.perf_counter()
            reg.setInputTargetDevice(dev[ti].data_ptr(), len(scans[ti]))
            reg.setInputSourceDevice(dev[si].data_ptr(), len(scans[si]))
            reg.align(guess)
            lat.append(time.perf_counter() - t1)
        single_ms = 1e3 * float(np.median(lat[1:]))
    # mean valid neighbour voxels per point of the evaluations actually launched (k-bar of SURVEY.md §8d); evaluations a
    # controller answers from its cache (repeated line-search trials) are neither launched nor counted
    kbar = float(launched[1] / launched[0]) if launched[0] else 0.0
    # ---- CPU baseline + parity on a bounded sample of the same pairs ---------------------------------------------------
    cpu = None
    parity = None
    if not args.no_cpu and args.cpu_pairs > 0:
        from oracle import oracle as orc

        host_cores = os.cpu_count() or 1
        ncpu = min(args.cpu_pairs, len(pairs))
        # thread count: the reference default (reg_num_threads: 8, config/mrg_slam.yaml:101) and wider settings up to the
        # host's cores; the fastest one is reported (the per-point OpenMP loop stops scaling well before 256 threads)
        sweep = sorted({t for t in (8, 16, 32, 64, host_cores) if t <= host_cores})

        def run_cpu(nt, sample):
            o = orc.Ndt(resolution=1.0, transformation_epsilon=args.eps, maximum_iterations=64, num_threads=nt)
            tc, out = time.perf_counter(), []
            for (ti, si, guess, _) in sample:
                o.setInputTarget(scans[ti])
                o.setInputSource(scans[si])
                o.align(guess)
                out.append((o.getFinalTransformation(), o.hasConverged(), o.getFinalNumIteration(), o.evals))
            return time.perf_counter() - tc, out

        probe = pairs[:min(4, ncpu)]  # thread-count probe on a few pairs, then the whole sample with the fastest setting
        cores = min(sweep, key=lambda nt: run_cpu(nt, probe)[0])
        tc, o_res = run_cpu(cores, pairs[:ncpu])
        cpu = {"value": ncpu / tc, "unit": "alignments/s", "cores": cores, "kind": "port",
               "sample": f"{ncpu} of the {args.batch} pairs of one step (setInputTarget+align each), CPU oracle = restated pclomp NDT_OMP, "
                         f"-O3 -fopenmp, fastest of OpenMP thread counts {sweep} on a {host_cores}-thread host = {cores} threads, {tc:.2f} s"}
        dts, drs, same = [], [], True
        for k in range(ncpu):
            Tg = result_matrix(res[k])
            To, conv, it, ev = o_res[k]
            dts.append(float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])))
            drs.append(rot_angle(Tg[:3, :3], To[:3, :3]))
            same = same and bool(res[k]["converged"]) == conv and int(res[k]["iterations"]) == it
        parity = {"pairs": ncpu, "max_dt_m": max(dts), "max_dr_rad": max(drs), "same_iterations_and_convergence": same, "bar": "1e-4 m / 1e-4 rad"}

    # dominant kernel = ndt_derivatives_kernel<0,7> (score + gradient + Hessian); the other two variants are listed beside it
    k_ms, k_launch, k_bytes = per_mode[0]
    achieved = (k_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
    variants = {name: {"launches": int(per_mode[m][1]), "avg_launch_ms": (per_mode[m][0] / per_mode[m][1]) if per_mode[m][1] else None,
                       "achieved_GBps": (per_mode[m][2] / 1e9) / (per_mode[m][0] / 1e3) if per_mode[m][0] > 0 else None}
                for m, name in enumerate(("ndt_derivatives_kernel<0,7>", "ndt_derivatives_kernel<1,7>", "ndt_derivatives_kernel<2,7>"))}
    # HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately with
    # rocprofv3 --pmc and corrected as MI355X_MICROARCH.md prescribes; profiles/summarize.py) - null until a profile exists
    traffic = None
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.json"))
        if prof:
            traffic = json.load(open(os.path.join(ROOT, "profiles", prof[-1]))).get("traffic_bytes_per_mean_launch")
    except OSError:
        pass
    true_err = float(np.mean([np.linalg.norm(result_matrix(res[b])[:3, 3] - pairs[b][3][:3, 3]) for b in range(args.batch)]))
    out = {
        "metric": "scan-pair alignments/sec (NDT, ~120k pts, 1.0 m voxel)",
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
                        f"{args.batch} pairs per GPU per step ({args.distinct} distinct pairs x warm guesses), mean {n_pts:.0f} pts/scan "
                        f"({'distance filter 0.1-35 m' if args.prefilter == 'distance' else 'distance + 0.1 m voxel + radius outlier prefilter'}), "
                        f"setInputTarget + setInputSource + align per pair, inputs resident in HBM",
            "pairs_per_gpu_per_step": args.batch,
            "points_per_scan": n_pts,
            "parallelism": f"{world} x 1 GPU, pairs sharded per rank, RCCL all-gather of 384-byte result records" if world > 1 else "1 GPU",
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel": "ndt_derivatives_kernel<0,7>", "avg_launch_ms": (k_ms / k_launch) if k_launch else None, "launches": int(k_launch),
                     "alg_bytes_per_launch": (k_bytes / k_launch) if k_launch else None,
                     "byte_model": "per launch: sum over active pairs of N_src*(16 + 7*8) + valid_neighbours*48 (SURVEY.md §8d)",
                     "variants": variants},
        "cpu_baseline": cpu,
        "parity_vs_oracle": parity,
        "evaluations_per_alignment": evals / (args.batch * args.steps),
        "iterations_per_alignment": iters / (args.batch * args.steps),
        "mean_valid_neighbours": kbar,
        "single_pair_latency_ms": single_ms,
        "mean_translation_error_vs_truth_m": true_err,
        "input_generation_s": t_gen,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
