cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
timeout -k 10 400 python3 -m pytest tests/test_gpu_fitness_passes.py tests/test_gpu_batch.py tests/test_gpu_filters.py -q -x -m gpu 2>&1 | tail -3 || exit 1
for i in 1 2; do python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8', d['ms_per_step'], d['config3_shard']['records_sha256_16'], d['roofline_fitness']['pyramid_walk_ms_per_step'])"; done
python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('g1', d['ms_per_step'], d['config3_shard']['records_sha256_16'])"
