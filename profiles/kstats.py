"""Prints per-kernel totals of a rocprofv3 --kernel-trace csv, divided by a step count:  python3 profiles/kstats.py <dir> <steps>"""
import csv, glob, sys
from collections import defaultdict

d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
tot, cnt = defaultdict(float), defaultdict(int)
t0, t1 = 1 << 62, 0
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot[k] += (e - b) / 1e6
    cnt[k] += 1
    t0, t1 = min(t0, b), max(t1, e)
print(f"{sum(tot.values()) / steps:9.3f} ms of kernels per step over {steps:g} steps")
for k in sorted(tot, key=tot.get, reverse=True)[:28]:
    print(f"{tot[k] / steps:9.3f} ms/step {cnt[k] / steps:8.1f} calls/step  {k}")
