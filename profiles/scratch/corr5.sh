cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_gicp.py tests/test_gpu_fitness_passes.py tests/test_gpu_filters.py tests/test_gpu_batch.py -q -x -m gpu 2>&1 | tail -4 || exit 1
for m in 1 2; do echo passes $m; MRGFE_GICP_CORR_PASSES=$m python3 profiles/gicp_profile.py frame130 2>/dev/null | tail -1; MRGFE_GICP_CORR_PASSES=$m python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1; done
python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1
python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8', d['ms_per_step'], d['config3_shard']['records_sha256_16'])"
