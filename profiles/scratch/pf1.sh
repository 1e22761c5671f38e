cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
for L in libmrgfe.so libmrgfe_pf.so libmrgfe.so libmrgfe_pf.so; do MRGFE_LIB=$GRAFT_REPO_ROOT/mrg_slam_amd/$L python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline_fitness']; print('$L', round(d['ms_per_step'],3), 'block', round(f['block_pass_ms_per_step'],3), 'sweep', round(f['ms_per_step'],3), d['config3_shard']['records_sha256_16'])"; done
