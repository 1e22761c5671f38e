#!/bin/bash
# Round-6 profile collection (run on the GPU box through gpurun from the repo root, one call per part: each stays under gpurun's 1200 s):
#   gpurun --timeout 1150 -- "bash profiles/collect_r06.sh trace r05 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r06.sh pmc r05 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r06.sh soak r05 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r06.sh soak2 r05 $(git rev-parse --short HEAD)"
#   gpurun --timeout 1150 -- "bash profiles/collect_r06.sh parity r05 $(git rev-parse --short HEAD)"
# then, back in the container:  python profiles/summarize.py r05
# As in round 4 every traced command launches the dominant kernel of its workload only in steps of that workload.  New in round 5: the headline's
# timed region keeps TWO batches in flight (bench.py --in-flight 2, mrgfe_batch_align_async), so config[1] is traced twice — one step at a time
# (--in-flight 1: the shape of rounds 1-4, the kernel with nothing beside it) and pipelined (--in-flight 2 --no-seq: the driver line's timed region);
# the PCL_NDT_HIP batch (registration_method "NDT") gets a trace and counter passes of its own; the soak covers it too.
part=${1:-trace}
tag=${2:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
git_rev=${3:-unknown}
python3 bench.py --prepare-only && python3 bench.py --mode shard --prepare-only || exit 1
C1="python3 bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --in-flight 1"
C1P="python3 bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --in-flight 2 --no-seq --steps 10"
C3="python3 bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 6 --warmup 2"
C38="python3 bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3"
GB="python3 profiles/gicp_profile.py batch"
PN="python3 profiles/pclndt_profile.py 64 1e-5 3"
if [ "$part" = trace ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- $C1 > gpurun_out/prof_$tag.log 2> gpurun_out/prof_$tag.err || exit 1
    echo "config[1] trace (one step at a time) done"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe_$tag -o s -- $C1P > gpurun_out/prof_pipe_$tag.log 2> gpurun_out/prof_pipe_$tag.err || exit 1
    echo "config[1] trace (two steps in flight) done"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard_$tag -o s -- $C3 > gpurun_out/prof_shard_$tag.log 2> gpurun_out/prof_shard_$tag.err || exit 1
    echo "config[3] trace done"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard8_$tag -o s -- $C38 > gpurun_out/prof_shard8_$tag.log 2> gpurun_out/prof_shard8_$tag.err || exit 1
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pclndt_$tag -o s -- $PN > gpurun_out/prof_pclndt_$tag.log 2> gpurun_out/prof_pclndt_$tag.err || exit 1
    echo "PCL_NDT_HIP trace done"
    for w in batch frame; do
        rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gicp_$w -o s -- python3 profiles/gicp_profile.py $w > gpurun_out/prof_gicp_$w.log 2>&1 || exit 1
        python3 profiles/gicp_profile.py $w 2>/dev/null | tail -1 > gpurun_out/gicp_$w.txt
    done
    echo "GICP traces done"
    python3 tests/extra_measurements.py > gpurun_out/extra_$tag.json 2> gpurun_out/extra_$tag.err
    echo "extra measurements done"
    python3 bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --steps 20 2>/dev/null | tail -1 > gpurun_out/bench_$tag.json
    $C3 2>/dev/null | tail -1 > gpurun_out/bench_shard_$tag.json
    $C38 2>/dev/null | tail -1 > gpurun_out/bench_shard8_$tag.json
    : > gpurun_out/inproc_$tag.jsonl
    for g in 1 2 4 8; do python3 bench.py --full-line --mode shard --inproc --gpus $g --steps 6 --warmup 2 2>/dev/null | tail -1 >> gpurun_out/inproc_$tag.jsonl; done
    for b in 32 64 128; do python3 bench.py --full-line --no-latency --no-cpu --no-extras --shard-steps 0 --batch $b --steps 10 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_b$b.json; done
elif [ "$part" = parity ]; then
    # the stated-size parity sweeps beyond what bench.py's line holds, with the round-5 library (new sort kernels under every grid; PCL_NDT_HIP in the sweep)
    python3 profiles/config1_parity_ranks.py 1,2,3,4,5,6,7 2> gpurun_out/config1_parity_ranks_$tag.err | tail -1 > gpurun_out/config1_parity_ranks_$tag.json
    echo "config[1] ranks done"
    python3 profiles/ndt_fullsize_sweep.py 64 2> gpurun_out/ndt_fullsize_sweep_$tag.err | tail -1 > gpurun_out/ndt_fullsize_sweep_$tag.json
    echo "full-size sweep done"
    python3 profiles/config2_parity.py 96 2> gpurun_out/config2_parity_$tag.err | tail -1 > gpurun_out/config2_parity_$tag.json
    echo "parity done"
elif [ "$part" = soak2 ]; then
    # the prefilter rows and the rows around the alignment, randomised (the radix sort under their grids changed in round 5)
    python3 profiles/soak_filters.py 3000 2> gpurun_out/soak_filters_$tag.err | tail -1 > gpurun_out/soak_filters_$tag.json
    python3 profiles/soak_misc.py 3000 2> gpurun_out/soak_misc_$tag.err | tail -1 > gpurun_out/soak_misc_$tag.json
    echo "filter / misc soaks done"
    python3 profiles/loop_parity_seeds.py $(python3 -c "print(','.join(str(4251 + i) for i in range(16)))") 2> gpurun_out/loop_parity_seeds2_$tag.err | tail -1 > gpurun_out/loop_parity_seeds2_$tag.json
    echo "soak2 done"
elif [ "$part" = soak ]; then
    python3 profiles/soak.py 3000 900 500 2000 > gpurun_out/soak_$tag.json 2> gpurun_out/soak_$tag.err
    echo "registration soak done"
    python3 profiles/loop_parity_seeds.py $(python3 -c "print(','.join(str(4243 + i) for i in range(8)))") 2> gpurun_out/loop_parity_seeds_$tag.err | tail -1 > gpurun_out/loop_parity_seeds_$tag.json
    echo "soak done"
else
    P1="$C1 --steps 1 --warmup 0"
    P3="python3 bench.py --full-line --no-latency --mode shard --no-cpu --no-extras --steps 1 --warmup 1"
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
    PN1="python3 profiles/pclndt_profile.py 64 1e-5 1"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_pclndt -o f -- $PN1 > gpurun_out/pmc_fetch_pclndt.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_pclndt -o w -- $PN1 > gpurun_out/pmc_write_pclndt.log 2>&1 || exit 1
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_sq_pclndt -o s -- $PN1 > gpurun_out/pmc_sq_pclndt.log 2>&1 || exit 1
    echo "PCL_NDT_HIP counters done"
fi
echo "$git_rev" > gpurun_out/collected_rev_$part.txt
date -u +%Y-%m-%dT%H:%MZ > gpurun_out/collected_date_$part.txt
echo "collect_r06 $part done"
