cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_gicp.py tests/test_gpu_primitives.py -q -x -m gpu 2>&1 | tail -5 || exit 1
for v in "4 8" "2 8" "2 16" "1 32" "4 4" "3 6"; do set -- $v; echo lanes $1 chunk $2; MRGFE_GICP_LANES=$1 MRGFE_GICP_CHUNK=$2 timeout -k 10 200 python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1 || exit 1; done
python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1
