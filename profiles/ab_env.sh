# A/B of an environment switch on the headline workload (config[1], 256 pairs: pipelined and one step at a time), config[3] and its shard of 8:
#   gpurun -- "bash profiles/ab_env.sh MRGFE_PLAN_IN_KERNEL 0 1"
var=$1; shift
python bench.py --prepare-only > /dev/null 2>&1; python bench.py --mode shard --prepare-only > /dev/null 2>&1
for i in 1 2; do for val in "$@"; do
  export $var=$val
  a=$(python bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', round(d['ms_per_step'],3), 'one at a time', round(d['value_one_step_at_a_time']['ms_per_step'],3), 'kernel ms', round(d['roofline']['one_step_at_a_time']['avg_launch_ms'],4))")
  b=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'])")
  c=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'])")
  echo "$var=$val: config1 $a | config3 $b | shard of 8 $c"
done; done
