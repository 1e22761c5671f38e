cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_filters.py tests/test_gpu_primitives.py -q -x -m gpu -k "knn or grid_set" 2>&1 | tail -2
STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20\|k 31\|k 10"; python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1
