cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>gpurun_out/set3.err | tail -1 > gpurun_out/set3.json
MRGFE_NO_EARLY_FIT=1 timeout -k 10 200 python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/set3_noearly.json
timeout -k 10 200 python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/set3_s8.json
