cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_gicp.py -q -x -m gpu 2>&1 | tail -6 || exit 1
python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1; python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1; python3 profiles/gicp_profile.py frame130 2>/dev/null | tail -1
