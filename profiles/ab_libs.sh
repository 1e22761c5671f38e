# A/B of library files on the headline workload (config[1], 256 pairs: pipelined and one step at a time), config[3] and the PCL_NDT_HIP batch:
#   gpurun -- "bash profiles/ab_libs.sh build/libmrgfe_x.so mrg_slam_amd/libmrgfe.so"
python bench.py --prepare-only > /dev/null 2>&1; python bench.py --mode shard --prepare-only > /dev/null 2>&1
for i in 1 2; do for lib in "$@"; do
  export MRGFE_LIB=$PWD/$lib MRGFE_LIB_ALLOW_MISSING=1
  a=$(python bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', round(d['ms_per_step'],3), 'one at a time', round(d['value_one_step_at_a_time']['ms_per_step'],3), 'kernel ms', round(d['roofline']['one_step_at_a_time']['avg_launch_ms'],4))")
  b=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'])")
  c=$(python profiles/pclndt_profile.py 64 1e-5 3 2>/dev/null | tail -1 | cut -c1-200)
  echo "$lib: config1 $a | config3 $b | pclndt $c"
done; done
