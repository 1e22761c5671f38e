# A/B of one environment switch of the fitness passes (AB_VAR over AB_VALUES) on config[3] (one GPU and a rank's shard of 8), after the GPU tests of the passes:
#   gpurun -- "AB_VAR=MRGFE_FIT_NEAR AB_VALUES='0.35 0 0.2 0.5' bash profiles/fit_ab.sh"
set -e
python -m pytest tests/test_gpu_fitness_passes.py tests/test_gpu_gicp.py tests/test_gpu_loop_detector.py -q -m gpu -x > gpurun_out/t_fit.log 2>&1 || { tail -30 gpurun_out/t_fit.log; exit 1; }
tail -2 gpurun_out/t_fit.log
python bench.py --full-line --no-latency --mode shard --prepare-only > /dev/null 2>&1
for v in ${AB_VALUES:-1 0 1 0}; do
  export ${AB_VAR:-MRGFE_FIT_NEAR}=$v
  b=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline_fitness']; print(round(d['ms_per_step'],3), d['config3_shard']['records_sha256_16'], 'fitness kernels ms', round(f['ms_per_step'],2), 'block', round(f['block_pass_ms_per_step'],2), 'points/query', round(f['candidate_points_per_queued_query'],1))")
  c=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --shard-of 8 --steps 16 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  echo "${AB_VAR:-MRGFE_FIT_NEAR}=$v config3 $b | shard8 $c"
done
