#!/bin/bash
# Round-3 additions to profiles/collect.sh (run it first): counter passes for the config[3] step (getFitnessScore kernels) and for the GICP
# batch / frame workloads, and their kernel traces.    gpurun --timeout 1200 -- 'bash profiles/collect_r03.sh r03'
# then, back in the container:  python profiles/summarize.py r03
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
S="python3 bench.py --mode shard --no-cpu --no-extras --steps 1 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_shard -o f -- $S > gpurun_out/pmc_fetch_shard.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_shard -o w -- $S > gpurun_out/pmc_write_shard.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq_shard -o s -- $S > gpurun_out/pmc_sq_shard.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard8_$tag -o s -- python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 6 --warmup 2 > gpurun_out/prof_shard8_$tag.log 2>&1
G="python3 profiles/gicp_profile.py batch"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_gicp -o f -- $G > gpurun_out/pmc_fetch_gicp.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_gicp -o w -- $G > gpurun_out/pmc_write_gicp.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq_gicp -o s -- $G > gpurun_out/pmc_sq_gicp.log 2>&1
for w in batch frame; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gicp_$w -o s -- python3 profiles/gicp_profile.py $w > gpurun_out/prof_gicp_$w.log 2>&1
    python3 profiles/gicp_profile.py $w 2>/dev/null | tail -1 > gpurun_out/gicp_$w.txt
done
python3 bench.py --mode shard --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_shard_$tag.json
python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_shard8_$tag.json
cat gpurun_out/gicp_batch.txt gpurun_out/gicp_frame.txt
