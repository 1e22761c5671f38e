cd $GRAFT_REPO_ROOT
( time python3 bench.py > gpurun_out/final_default.json 2> gpurun_out/final_default.err ) 2> gpurun_out/final_default.time
tail -3 gpurun_out/final_default.time
( time BENCH_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/final_g2.json 2> gpurun_out/final_g2.err ) 2> gpurun_out/final_g2.time
tail -3 gpurun_out/final_g2.time
tail -c 600 gpurun_out/final_g2.json
