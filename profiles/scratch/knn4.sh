cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_filters.py tests/test_gpu_primitives.py -q -x -m gpu -k "knn or grid_set" 2>&1 | tail -4 || exit 1
STATS=1 timeout -k 10 200 python3 profiles/scratch/knn1.py 2>&1 | tail -3 && STATS=0 timeout -k 10 200 python3 profiles/scratch/knn1.py 2>&1 | tail -3
