cd $GRAFT_REPO_ROOT
for L in libmrgfe.so libmrgfe_k7.so libmrgfe_k6.so libmrgfe.so; do echo $L; MRGFE_LIB=$GRAFT_REPO_ROOT/mrg_slam_amd/$L STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20\|k 31"; done
