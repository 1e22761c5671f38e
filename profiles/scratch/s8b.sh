cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
BENCH_STEP_PHASES=1 python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>&1 >/dev/null | grep "step phases"
BENCH_STEP_PHASES=1 python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>&1 >/dev/null | grep "step phases"
