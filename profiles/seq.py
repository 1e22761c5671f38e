"""Kernel sequence of the last step of a rocprofv3 kernel trace of bench.py (derivative rounds only): python3 profiles/seq.py <dir> [rounds]"""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")) for r in csv.DictReader(open(f)))
plans = [i for i, r in enumerate(rows) if r[2].startswith("ndt_plan")][-n:]
for a, b in zip(plans, plans[1:] + [len(rows)]):
    seg = [r for r in rows[a:b] if r[2].startswith("ndt_")][:6]
    print("  ".join(f"{r[2][4:24]:>20s} {(r[1] - r[0]) / 1e3:7.1f}" for r in seg), f"| round {(seg[-1][1] - seg[0][0]) / 1e3:7.1f} us")
