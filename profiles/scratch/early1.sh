cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
run() { env "$@" python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'])"; }
run X=1
run MRGFE_EARLY_FIT_MIN_PAIRS=8 MRGFE_EARLY_FIT_WAVE_PERCENT=50
run MRGFE_EARLY_FIT_MIN_PAIRS=8 MRGFE_EARLY_FIT_WAVE_PERCENT=70
run MRGFE_EARLY_FIT_MIN_PAIRS=8 MRGFE_EARLY_FIT_WAVE_PERCENT=85
run X=1
run2() { env "$@" python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G1 $*', round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'])"; }
run2 X=1
run2 MRGFE_EARLY_FIT_MIN_PAIRS=8 MRGFE_EARLY_FIT_WAVE_PERCENT=60
run2 MRGFE_EARLY_FIT_MIN_PAIRS=8 MRGFE_EARLY_FIT_WAVE_PERCENT=80
