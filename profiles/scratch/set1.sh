cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_primitives.py -q -x -m gpu -k "grid_set" 2>&1 | tail -15 && timeout -k 10 400 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fitness_passes.py -q -x -m gpu 2>&1 | tail -5 && python3 bench.py --no-extras --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('records_sha256_16'), d.get('config',{}).get('workload'))" && for v in "2 16" "1 16" "1 64" "2 8" "4 4"; do set -- $v; echo builders $1 chunk $2; MRGFE_FIT_BUILDERS=$1 MRGFE_FIT_CHUNK=$2 python3 bench.py --no-extras --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('records_sha256_16'))"; done && python3 bench.py --no-extras --shard-of 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8', d['ms_per_step'], d.get('records_sha256_16'))"
