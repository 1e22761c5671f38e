cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
for v in default "MRGFE_EARLY_FIT_MIN_PAIRS=8" "MRGFE_FIT_BUILDERS=1" "MRGFE_FIT_CHUNK=4"; do
  echo $v; if [ "$v" = default ]; then e=X=1; else e=$v; fi
  env $e python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8', d['ms_per_step'], d['config3_shard']['records_sha256_16'])"
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_s8_trace -o s -- python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 6 --warmup 2 > gpurun_out/r3_s8_trace.log 2>&1
python3 profiles/shard_timeline.py gpurun_out/r3_s8_trace
