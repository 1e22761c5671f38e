cd $GRAFT_REPO_ROOT
for c in adaptive 1.0 0.5 0.25 0.125; do echo cell $c; if [ $c = adaptive ]; then STATS=1 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20"; STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20"; else MRGFE_NN_CELL=$c STATS=1 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20"; MRGFE_NN_CELL=$c STATS=0 python3 profiles/scratch/knn1.py 2>&1 | grep "k 20"; fi; done
