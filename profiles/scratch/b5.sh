cd $GRAFT_REPO_ROOT
python3 bench.py --prepare-only > /dev/null 2>&1
for m in 1 0 1 0; do MRGFE_PREP_OVERLAP=$m python3 bench.py --no-cpu --no-extras --shard-steps 0 --steps 12 --warmup 3 --latency 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap $m', d['value'], d['ms_per_step'], d.get('single_pair_latency_ms'))"; done
