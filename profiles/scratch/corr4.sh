cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MRGFE_GICP_CORR_PASSES=2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_corr_single -o s -- python3 profiles/gicp_profile.py frame130 > gpurun_out/r3_corr_single.log 2>&1
python3 profiles/kstats.py gpurun_out/r3_corr_single 1 | head -16
