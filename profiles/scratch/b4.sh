cd $GRAFT_REPO_ROOT
python3 bench.py --prepare-only > /dev/null 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_ndt.py tests/test_gpu_batch.py tests/test_gpu_control.py tests/test_gpu_configs.py -q -x -m gpu 2>&1 | tail -3
for i in 1 2 3; do python3 bench.py --no-cpu --no-extras --shard-steps 0 --steps 10 --warmup 3 --latency 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('single_pair_latency_ms'))"; done
