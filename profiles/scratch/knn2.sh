cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STATS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_knn_trace -o s -- python3 profiles/scratch/knn1.py > gpurun_out/r3_knn_trace.log 2>&1
python3 profiles/kstats.py gpurun_out/r3_knn_trace 1 | head -12
