cd $GRAFT_REPO_ROOT
( time BENCH_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/final_g2.json 2> gpurun_out/final_g2.err ) 2> gpurun_out/final_g2.time
tail -3 gpurun_out/final_g2.time
python3 -c "
import json
d=json.loads(open('gpurun_out/final_g2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','cpu_baseline','parity_vs_oracle','value_host_pointers')}, d['config3_shard']['records_sha256_16'])"
