cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
for i in 1 2; do BENCH_STEP_PHASES=1 python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>gpurun_out/s8d.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config3_shard']; print('shard8', d['ms_per_step'], c['records_sha256_16'], c['matched_keyframes'])"; grep "step phases" gpurun_out/s8d.err; done
python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config3_shard']; print('g1', d['ms_per_step'], c['records_sha256_16'], c['matched_keyframes'])"
