cd $GRAFT_REPO_ROOT
for L in libmrgfe.so libmrgfe_s6.so libmrgfe_s9.so libmrgfe_s16.so libmrgfe_s24.so; do echo $L; MRGFE_LIB=$GRAFT_REPO_ROOT/mrg_slam_amd/$L STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20\|k 31\|k 10" | cut -c1-80; done
