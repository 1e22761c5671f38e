"""One rank of a two-rank loop-closure candidate matching on ONE GPU: the real BatchMatcher on this rank's shard, the 384-byte records
all-gathered over gloo (RCCL refuses two ranks on one device; the gather code is the same), the best-candidate replay on every rank
(mrg_slam_amd/loop_closure.py, /root/reference/src/mrg_slam/loop_detector.cpp:104,126-145).

    python tests/workers/gloo_matcher_worker.py <rank> <world> <port> <n_candidates> <out.npy> [backend]

backend "nccl" with world 1: the RCCL process group of a ONE-rank job — init, all_gather_into_tensor on device tensors (loop_closure.py
gather_records' `use_cuda` branch) and teardown run on the one-GPU box (a port other than 0 asks for the process group).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def workload(n):
    from conftest import small_cloud
    from mrg_slam_amd import synth

    rng = np.random.default_rng(99)
    tgt = small_cloud(20000, 700, extent=(30.0, 20.0, 3.0))
    cands, guesses = [], []
    for k in range(n):
        rel = synth.make_pose(rng.normal(0, 0.3, 3) * [1, 1, 0.1], synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        src = (np.linalg.inv(rel) @ np.c_[tgt[: 9000 + 37 * k, :3], np.ones(9000 + 37 * k)].T).T.astype(np.float32)
        src[:, 3] = tgt[: len(src), 3]
        cands.append(np.ascontiguousarray(src))
        guesses.append(synth.perturb_pose(np.eye(4), rng))
    return tgt, cands, guesses


def main():
    rank, world, port, n, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    use_pg = world > 1 or port != 0
    import torch  # noqa: F401  (before libmrgfe)
    import torch.distributed as dist

    from mrg_slam_amd import BatchMatcher
    from mrg_slam_amd import loop_closure as lc

    if use_pg:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    tgt, cands, guesses = workload(n)
    rec, best, score = lc.match_candidates(lambda: BatchMatcher(transformation_epsilon=0.01, maximum_iterations=64), tgt, cands, guesses)
    if rank == 0:
        np.save(out, rec)
        if use_pg:
            print("backend", dist.get_backend(), "world", dist.get_world_size())
        print(best, score)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
