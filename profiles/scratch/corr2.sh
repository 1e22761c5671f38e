cd $GRAFT_REPO_ROOT
for m in 0 1; do echo passes $m; MRGFE_GICP_CORR_PASSES=$m python3 profiles/gicp_profile.py frame130 2>/dev/null | tail -1; done
