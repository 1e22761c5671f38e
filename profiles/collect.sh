#!/bin/bash
# Collects the rocprofv3 inputs of profiles/summarize.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r02'
# then, back in the container:  python profiles/summarize.py r02
# Counter passes are separate runs and carry no tracing domain other than the kernel trace (gpurun refuses other mixes).
# The synthetic scans are generated (forked workers) and cached BEFORE any profiled run: under rocprofv3 the GPU is initialised
# before the program starts, and such a process must neither fork nor exec.
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py --prepare-only
python3 bench.py --latency > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --no-cpu --shard-steps 0 > gpurun_out/prof_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu --shard-steps 0 --steps 1 --warmup 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --no-cpu --shard-steps 0 --steps 1 --warmup 0 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o s -- python3 bench.py --no-cpu --shard-steps 0 --steps 1 --warmup 0 > gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard_$tag -o s -- python3 bench.py --mode shard --steps 3 --warmup 1 > gpurun_out/prof_shard_$tag.log 2>&1
for w in gicp gicp_full prefilter fitness lc gicp_lc; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_side_$w -o s -- python3 profiles/side_workloads.py $w > gpurun_out/side_$w.log 2>&1
done
python3 tests/extra_measurements.py > gpurun_out/extra_$tag.json 2> gpurun_out/extra_$tag.err
for b in 32 64 128; do python3 bench.py --no-cpu --shard-steps 0 --batch $b --steps 10 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_b$b.json; done
tail -1 gpurun_out/bench_$tag.json | cut -c1-400
