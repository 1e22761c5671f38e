#!/bin/bash
# Round-4 profile collection (run on the GPU box through gpurun from the repo root, four calls: each stays under gpurun's 1200 s):
#   gpurun --timeout 1150 -- "bash profiles/collect_r04.sh trace r04 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r04.sh pmc r04 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r04.sh soak r04 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r04.sh parity r04 $(git rev-parse --short HEAD)"
# then, back in the container:  python profiles/summarize.py r04
# Every traced command launches the dominant kernel of its workload ONLY in steps of that workload (no CPU legs, no side measurements, no
# config[3] leg inside the config[1] run), so the per-kernel average of `--kernel-trace --stats` is over the launch population bench.py's own
# HIP events time, and profiles/summarize.py can recompute `roofline.frac` from the trace alone.
# Counter passes are separate runs and carry no tracing domain other than the kernel trace (gpurun refuses other mixes); the program
# follows `--` directly.  The synthetic scans are generated (forked workers) and cached BEFORE any profiled run: under rocprofv3 the GPU
# is initialised before the program starts, and such a process must neither fork nor exec.
part=${1:-trace}
tag=${2:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
git_rev=${3:-unknown}   # the snapshot on the GPU box has no .git: the caller passes `git rev-parse --short HEAD`
python3 bench.py --prepare-only && python3 bench.py --mode shard --prepare-only || exit 1
C1="python3 bench.py --no-cpu --no-extras --shard-steps 0"
C3="python3 bench.py --mode shard --no-cpu --no-extras --steps 6 --warmup 2"
C38="python3 bench.py --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3"
GB="python3 profiles/gicp_profile.py batch"
if [ "$part" = trace ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- $C1 > gpurun_out/prof_$tag.log 2> gpurun_out/prof_$tag.err || exit 1
    echo "config[1] trace done"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard_$tag -o s -- $C3 > gpurun_out/prof_shard_$tag.log 2> gpurun_out/prof_shard_$tag.err || exit 1
    echo "config[3] trace done"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard8_$tag -o s -- $C38 > gpurun_out/prof_shard8_$tag.log 2> gpurun_out/prof_shard8_$tag.err || exit 1
    for w in batch frame; do
        rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gicp_$w -o s -- python3 profiles/gicp_profile.py $w > gpurun_out/prof_gicp_$w.log 2>&1 || exit 1
        python3 profiles/gicp_profile.py $w 2>/dev/null | tail -1 > gpurun_out/gicp_$w.txt
    done
    echo "GICP traces done"
    for w in prefilter lc gicp_lc; do
        rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_side_$w -o s -- python3 profiles/side_workloads.py $w > gpurun_out/side_$w.log 2>&1 || exit 1
    done
    python3 profiles/overlap_steps.py 12 2>/dev/null | tail -1 > gpurun_out/overlap_$tag.json
    python3 tests/extra_measurements.py > gpurun_out/extra_$tag.json 2> gpurun_out/extra_$tag.err
    echo "extra measurements done"
    $C3 2>/dev/null | tail -1 > gpurun_out/bench_shard_$tag.json
    $C38 2>/dev/null | tail -1 > gpurun_out/bench_shard8_$tag.json
    for b in 32 64 128; do python3 bench.py --no-cpu --no-extras --shard-steps 0 --batch $b --steps 10 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_b$b.json; done
elif [ "$part" = soak ]; then
    # the randomised parity runs (about 13 minutes): registration methods, prefilter rows, the rows around the alignment, config[3] for 24 more draws
    python3 profiles/soak.py 10000 3000 > gpurun_out/soak_$tag.json 2> gpurun_out/soak_$tag.err
    echo "registration soak done"
    python3 profiles/soak_filters.py 3000 2> gpurun_out/soak_filters_$tag.err | tail -1 > gpurun_out/soak_filters_$tag.json
    python3 profiles/soak_misc.py 3000 2> gpurun_out/soak_misc_$tag.err | tail -1 > gpurun_out/soak_misc_$tag.json
    echo "filter / misc soaks done"
    python3 profiles/loop_parity_seeds.py $(python3 -c "print(','.join(str(4243 + i) for i in range(24)))") 2> gpurun_out/loop_parity_seeds_$tag.err | tail -1 > gpurun_out/loop_parity_seeds_$tag.json
    echo "soak done"
elif [ "$part" = parity ]; then
    # the stated-size parity sweeps beyond what bench.py's line holds (about 8 minutes, most of it ray-casting the other ranks' streets)
    python3 profiles/config1_parity_ranks.py 1,2,3,4,5,6,7 2> gpurun_out/config1_parity_ranks_$tag.err | tail -1 > gpurun_out/config1_parity_ranks_$tag.json
    echo "config[1] ranks done"
    python3 profiles/ndt_fullsize_sweep.py 64 2> gpurun_out/ndt_fullsize_sweep_$tag.err | tail -1 > gpurun_out/ndt_fullsize_sweep_$tag.json
    python3 profiles/config2_parity.py 96 2> gpurun_out/config2_parity_$tag.err | tail -1 > gpurun_out/config2_parity_$tag.json
    echo "parity done"
else
    P1="$C1 --steps 1 --warmup 0"
    P3="python3 bench.py --mode shard --no-cpu --no-extras --steps 1 --warmup 1"
    SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- $P1 > gpurun_out/pmc_fetch.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- $P1 > gpurun_out/pmc_write.log 2>&1 || exit 1
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o s -- $P1 > gpurun_out/pmc_sq.log 2>&1 || exit 1
    echo "config[1] counters done"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_shard -o f -- $P3 > gpurun_out/pmc_fetch_shard.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_shard -o w -- $P3 > gpurun_out/pmc_write_shard.log 2>&1 || exit 1
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_sq_shard -o s -- $P3 > gpurun_out/pmc_sq_shard.log 2>&1 || exit 1
    echo "config[3] counters done"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_gicp -o f -- $GB > gpurun_out/pmc_fetch_gicp.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_gicp -o w -- $GB > gpurun_out/pmc_write_gicp.log 2>&1 || exit 1
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_sq_gicp -o s -- $GB > gpurun_out/pmc_sq_gicp.log 2>&1 || exit 1
fi
echo "$git_rev" > gpurun_out/collected_rev_$part.txt
date -u +%Y-%m-%dT%H:%MZ > gpurun_out/collected_date_$part.txt
echo "collect_r04 $part done"
