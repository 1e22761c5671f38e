cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_gicp.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_soak.py -q -x -m gpu 2>&1 | tail -3
for m in 1 0; do echo seed $m; MRGFE_GICP_CORR_SEED=$m python3 profiles/gicp_profile.py frame130 2>/dev/null | tail -1; MRGFE_GICP_CORR_SEED=$m python3 profiles/gicp_profile.py frame 2>/dev/null | tail -1; done
