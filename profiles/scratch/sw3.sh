cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
MRGFE_FIT_STATS=2 MRGFE_NO_EARLY_FIT=1 python3 bench.py --mode shard --no-cpu --no-extras --steps 2 --warmup 1 2>&1 >/dev/null | grep "fitness sweep" | tail -2
