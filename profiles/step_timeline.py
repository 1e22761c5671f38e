"""One timed step of `bench.py --mode shard` from a rocprofv3 kernel trace, as a timeline (python3 profiles/step_timeline.py <dir> [step index, default 8]):
steps are delimited by the target build's first kernel; per step the spans of the target build, the alignment rounds (with the per-round split
plan | derivative | controller and the gaps between them), the fitness grids and the fitness passes, in ms from the step's first kernel."""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")) for r in csv.DictReader(open(f)))
heads = [i for i, r in enumerate(rows) if r[2].startswith("bbox_partial") and (i == 0 or not rows[i - 1][2].startswith("bbox_"))]
# a step's build may run bbox_partial more than once: keep the heads that follow a fitness pass or start the trace
steps = [h for n, h in enumerate(heads) if n == 0 or any("nn_fit" in r[2] or "nn_fitness" in r[2] for r in rows[heads[n - 1]:h])]
a, b = steps[k], steps[k + 1]
seg = rows[a:b]
t0 = seg[0][0]
ms = lambda t: (t - t0) / 1e6


def span(pred, name):
    xs = [r for r in seg if pred(r[2])]
    if xs:
        print(f"  {name:18s} {ms(xs[0][0]):7.3f} .. {ms(max(x[1] for x in xs)):7.3f} ms, {sum(r[1] - r[0] for r in xs) / 1e6:6.3f} ms in {len(xs)} kernels")


print(f"step {k}: {ms(max(r[1] for r in seg)):.3f} ms from its first kernel to the end of its last")
span(lambda n: n.startswith(("bbox_", "ndt_cellkey", "rs_", "scan_", "ndt_segments", "ndt_big", "ndt_leaf", "ndt_dd")), "ndt target build")
span(lambda n: any(w in n for w in ("ndt_deriv", "ndt_reduce", "ndt_plan", "ndt_snapshot")), "alignment rounds")
span(lambda n: n.startswith("nn_") and "fit" not in n, "fitness grids")
span(lambda n: "nn_fit" in n, "fitness passes")
rk = [r for r in seg if any(w in r[2] for w in ("ndt_plan", "ndt_deriv", "ndt_reduce"))]
tot = [0.0] * 4
n_rounds = 0
i = 0
while i < len(rk):
    if rk[i][2].startswith("ndt_deriv"):
        d = rk[i]
        r = rk[i + 1] if i + 1 < len(rk) and rk[i + 1][2].startswith("ndt_reduce") else None
        tot[1] += d[1] - d[0]
        if r:
            tot[2] += r[1] - r[0]
        n_rounds += 1
    elif rk[i][2].startswith("ndt_plan"):
        tot[0] += rk[i][1] - rk[i][0]
    i += 1
wall = rk[-1][1] - rk[0][0]
print(f"  {n_rounds} rounds over {wall / 1e6:.3f} ms: plan {tot[0] / 1e6:.3f}, derivative {tot[1] / 1e6:.3f}, controller {tot[2] / 1e6:.3f}, gaps {(wall - sum(tot[:3])) / 1e6:.3f} ms")
for r in seg:
    if "nn_fit" in r[2] or "snapshot" in r[2]:
        print(f"    {ms(r[0]):7.3f} .. {ms(r[1]):7.3f}  {r[2]}")
