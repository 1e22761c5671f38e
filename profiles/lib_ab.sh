# A/B of two library files (MRGFE_LIB) on config[3] and the GICP batch: OLD=build/libmrgfe_old.so (default) against the in-tree library
#   gpurun -- "bash profiles/lib_ab.sh"
set -e
python -m pytest tests/test_gpu_fitness_passes.py tests/test_gpu_gicp.py tests/test_gpu_loop_detector.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/t_lib.log 2>&1 || { tail -30 gpurun_out/t_lib.log; exit 1; }
tail -2 gpurun_out/t_lib.log
python bench.py --full-line --no-latency --mode shard --prepare-only > /dev/null 2>&1
for i in 1 2; do for lib in ${OLD:-build/libmrgfe_old.so} mrg_slam_amd/libmrgfe.so; do
  export MRGFE_LIB=$PWD/$lib MRGFE_LIB_ALLOW_MISSING=1
  b=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline_fitness']; print(round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'], 'far pass ms', round(f['ms_per_step'],2), 'block', round(f['block_pass_ms_per_step'],2))")
  c=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --shard-of 8 --steps 16 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  g=$(python profiles/gicp_profile.py batch 2>/dev/null | tail -1)
  echo "$lib: config3 $b | shard8 $c | gicp $g"
done; done
