cd $GRAFT_REPO_ROOT
for v in "2 16" "1 16" "1 64" "2 8" "4 4" "3 8"; do set -- $v; echo builders $1 chunk $2; MRGFE_FIT_BUILDERS=$1 MRGFE_FIT_CHUNK=$2 timeout -k 10 200 python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('records_sha256_16'), d.get('inputs_sha256_16'))" || exit 1; done
timeout -k 10 200 python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8', d['ms_per_step'], d.get('records_sha256_16'))"
