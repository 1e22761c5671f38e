"""Per lock-step round of the last step in a rocprofv3 kernel trace of bench.py: duration (us) of the plan kernel, the three derivative
variants and the controller kernel, and the gaps between them:   python3 profiles/rounds.py <dir> [rounds]"""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")) for r in csv.DictReader(open(f)))
plans = [i for i, r in enumerate(rows) if r[2].startswith("ndt_plan")][-n:]
print("   plan     <0>     <1>     <2>  reduce |  gaps between them (us)")
tot = [0.0] * 5
for i in plans:
    seg = rows[i:i + 5]
    d = [(s[1] - s[0]) / 1e3 for s in seg]
    g = [(seg[k + 1][0] - seg[k][1]) / 1e3 for k in range(4)]
    tot = [a + b for a, b in zip(tot, d)]
    print(" ".join(f"{x:7.1f}" for x in d), "|", " ".join(f"{x:5.1f}" for x in g))
print(" ".join(f"{x:7.1f}" for x in tot), "| sum;  first plan to last reduce:", f"{(rows[plans[-1] + 4][1] - rows[plans[0]][0]) / 1e3:.1f} us")
