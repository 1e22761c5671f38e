cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in batch frame; do
  python3 profiles/gicp_profile.py $w 2>/dev/null | tail -1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_gicp_$w -o s -- python3 profiles/gicp_profile.py $w > gpurun_out/r3_gicp_$w.log 2>&1
  python3 profiles/kstats.py gpurun_out/r3_gicp_$w 1 | head -24
done
