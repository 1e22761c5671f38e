"""Timeline of the tail of a rocprofv3 kernel trace: python3 profiles/timeline.py <dir> [last_ms] — start offset, duration, gap to the previous
kernel's end (a negative gap = overlap with another stream), name; then totals of kernel time / idle gaps over the window."""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")) for r in csv.DictReader(open(f)))
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
t0, prev, busy, idle = win[0][0], win[0][0], 0, 0
for s, e, name in win:
    gap = s - prev
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {name[:70]}")
    busy += e - s
    idle += max(gap, 0)
    prev = max(prev, e)
print(f"window {(t_end - t0) / 1e3:.1f} us: kernels {busy / 1e3:.1f} us, idle gaps {idle / 1e3:.1f} us, {len(win)} launches")
