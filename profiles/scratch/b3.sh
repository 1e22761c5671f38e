cd $GRAFT_REPO_ROOT
python3 bench.py --prepare-only > /dev/null 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_ndt.py tests/test_gpu_filters.py tests/test_gpu_batch.py tests/test_golden.py tests/test_gpu_primitives.py tests/test_gpu_gicp.py -q -x -m gpu 2>&1 | tail -4
for i in 1 2; do python3 bench.py --no-cpu --shard-steps 0 --steps 10 --warmup 3 --latency 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('single_pair_latency_ms'), d['gpu_split_ms_per_step']['set_target_ms'])"; done
python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1; python3 profiles/gicp_profile.py frame130 2>/dev/null | tail -1; python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1
