"""Per-step anatomy from a rocprofv3 kernel trace of bench.py (python profiles/step_anatomy.py <trace.csv>): wall time of the last
steps, GPU busy time, idle share, rounds, and the busy time per kernel."""
import collections
import csv
import sys


def main(path, steps=3):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    bb = [i for i, e in enumerate(ev) if "bbox_partial" in e[2]]
    per = collections.Counter()
    for a, b in zip(bb[-steps - 1:-1], bb[-steps:]):
        seg = ev[a:b]
        t0 = seg[0][0]
        busy, (cs, ce) = 0, (seg[0][0], seg[0][1])
        for s, e, n in seg[1:]:
            if s <= ce:
                ce = max(ce, e)
            else:
                busy += ce - cs
                cs, ce = s, e
        busy += ce - cs
        wall = ev[b][0] - t0
        rounds = sum(1 for e in seg if "ndt_plan" in e[2] or "ndt_reduce" in e[2])
        print(f"step wall {wall / 1e6:.3f} ms, GPU busy {busy / 1e6:.3f} ms, idle {100 * (wall - busy) / wall:.1f} %, rounds {rounds}")
        for s, e, n in seg:
            per[n.split("(")[0].replace("void ", "").replace("mrgfe::", "")[:44]] += (e - s) / steps
    for k, v in per.most_common(14):
        print(f"  {v / 1e6:8.3f} ms  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
