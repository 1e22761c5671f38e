cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_fitness_passes.py tests/test_gpu_batch.py tests/test_gpu_primitives.py tests/test_gpu_filters.py -q -x -m gpu 2>&1 | tail -2
for i in 1 2; do python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline_fitness']; print('G1', round(d['ms_per_step'],3), 'block', round(f['block_pass_ms_per_step'],3), 'sweep', round(f['ms_per_step'],3), d['config3_shard']['records_sha256_16'])"; done
python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1; python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1
