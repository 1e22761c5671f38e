# A/B of one environment switch (AB_VAR, values 1 0 1 0) over the three NDT workloads, after the GPU tests named in AB_TESTS:
#   gpurun -- "AB_VAR=MRGFE_FUSED_PLAN bash profiles/ab.sh"
set -e
python -m pytest ${AB_TESTS:-tests/test_gpu_ndt.py tests/test_gpu_batch.py tests/test_gpu_soak.py tests/test_gpu_pclndt.py} -q -m gpu -x > gpurun_out/t_plan.log 2>&1 || { tail -30 gpurun_out/t_plan.log; exit 1; }
tail -3 gpurun_out/t_plan.log
python bench.py --full-line --no-latency --prepare-only >/dev/null 2>&1; python bench.py --full-line --no-latency --mode shard --prepare-only > /dev/null 2>&1
for v in 1 0 1 0; do
  export ${AB_VAR:-MRGFE_FUSED_PLAN}=$v
  a=$(python bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value_one_step_at_a_time']['ms_per_step'],3))")
  b=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d.get('config',{}).get('records_sha256_16'))")
  c=$(python bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --shard-of 8 --steps 16 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  echo "${AB_VAR:-MRGFE_FUSED_PLAN}=$v headline(pipe,seq) $a | config3 $b | shard8 $c"
done
