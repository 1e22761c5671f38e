cd $GRAFT_REPO_ROOT
for L in libmrgfe.so libmrgfe_k7.so libmrgfe_k8.so libmrgfe.so; do echo $L; MRGFE_LIB=$GRAFT_REPO_ROOT/mrg_slam_amd/$L STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20\|k 31"; MRGFE_LIB=$GRAFT_REPO_ROOT/mrg_slam_amd/$L python3 profiles/gicp_profile.py batch 2>/dev/null | tail -1; done
