cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gb -o s -- python3 tests/extra_measurements.py --gicp-batch-only > gpurun_out/prof_gb.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_gb/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]: print(r['Name'][:50], r['Calls'], round(float(r['TotalDurationNs'])/1e3,1), round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
