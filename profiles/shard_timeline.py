"""Phases of a config[3] step from a rocprofv3 kernel trace of `bench.py --mode shard`:  python3 profiles/shard_timeline.py <dir>
Per step (delimited by nn_fit_block_kernel launches): when the NDT target build, the derivative rounds, the fitness-grid builds and the
fitness passes start and end (ms from the step's first kernel), their summed kernel time and launch count."""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")) for r in csv.DictReader(open(f)))
fits = [i for i, r in enumerate(rows) if r[2].startswith("nn_fit_block")]
for s in range(max(1, len(fits) - 3), len(fits)):
    prev_end = max(r[1] for r in rows[fits[s - 1]:fits[s - 1] + 4] if "nn_fit" in r[2])
    this_end = max(r[1] for r in rows[fits[s]:fits[s] + 4] if "nn_fit" in r[2])
    seg = [r for r in rows if prev_end <= r[0] <= this_end]
    t0 = seg[0][0]

    def span(pred):
        xs = [r for r in seg if pred(r[2])]
        return f"{(xs[0][0] - t0) / 1e6:7.2f} .. {(xs[-1][1] - t0) / 1e6:7.2f} ms, {sum(r[1] - r[0] for r in xs) / 1e6:6.2f} ms in {len(xs)} kernels" if xs else "-"

    print(f"step {s}: {(this_end - t0) / 1e6:.2f} ms")
    print("  ndt target build :", span(lambda k: k.startswith("ndt_") and not any(w in k for w in ("deriv", "reduce", "plan"))))
    print("  derivative rounds:", span(lambda k: any(w in k for w in ("deriv", "reduce", "plan"))))
    print("  fitness grids    :", span(lambda k: k.startswith("nn_") and "fit" not in k))
    print("  fitness passes   :", span(lambda k: "nn_fit" in k))
