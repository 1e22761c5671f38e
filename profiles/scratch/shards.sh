cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
for g in 0 2 4 8; do
  a=""; if [ $g != 0 ]; then a="--shard-of $g"; fi
  python3 bench.py --mode shard --no-cpu --no-extras $a --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config3_shard']; print('G', '$g', round(d['ms_per_step'],3), c['pairs_per_gpu'], c['targets_built_per_gpu'], c['records_sha256_16'], round(d['roofline']['ms_per_step'],2), round(d['roofline_fitness']['ms_per_step'],2), round(d['roofline_fitness']['block_pass_ms_per_step'],2))"
done
