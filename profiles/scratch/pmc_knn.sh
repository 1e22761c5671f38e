cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STATS=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r3_pmc_knn -o s -- python3 profiles/scratch/knn1.py > gpurun_out/r3_pmc_knn.log 2>&1
STATS=0 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/r3_pmc_knn2 -o s -- python3 profiles/scratch/knn1.py > gpurun_out/r3_pmc_knn2.log 2>&1
ls gpurun_out/r3_pmc_knn gpurun_out/r3_pmc_knn2
