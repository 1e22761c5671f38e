# target-build check: GPU tests that cover the sort and the leaves, then the build alone timed and traced (profiles/build_profile.py)
set -e
python -m pytest tests/test_gpu_ndt.py tests/test_gpu_primitives.py tests/test_gpu_filters.py tests/test_gpu_mapcloud.py tests/test_gpu_fitness_passes.py -q -m gpu -x > gpurun_out/t_build.log 2>&1 || { tail -30 gpurun_out/t_build.log; exit 1; }
tail -2 gpurun_out/t_build.log
python profiles/build_profile.py 256 10 | tail -1
python profiles/build_profile.py 1 20 | tail -1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_build -o b -- python3 profiles/build_profile.py 256 10 > gpurun_out/prof_build.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_build/b_kernel_stats.csv")))
for r in rows[:14]: print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
