#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that a profiling `gpurun` call merged into gpurun_out/ into the small, tracked summaries
under profiles/ (gpurun_out/ is scratch and ignored by git).

    python profiles/summarize.py r01            # reads gpurun_out/{prof,pmc_fetch,pmc_write,pmc_sq}_*, writes profiles/r01_*

Commands that produced the inputs (run on the GPU box from /tmp with TMPDIR=/tmp, program directly after `--`):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o r01 -- python3 bench.py --no-cpu
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu --steps 1 --warmup 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --no-cpu --steps 1 --warmup 0
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace \
              --output-format csv -d gpurun_out/pmc_sq -o s -- python3 bench.py --no-cpu --steps 1 --warmup 0
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate passes; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled (an upper estimate for this kernel, whose
reads are a mix of 16-byte streams and scattered 48-byte records).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
DOMINANT = "ndt_derivatives_all_kernel<7>"  # one launch per round: the work items of all three evaluation kinds


def short(name):
    name = name.replace("void mrgfe::", "").replace("mrgfe::", "")
    return name.split("(")[0]


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    lines = [f"# rocprofv3 summary {tag} (MI355X, gfx950) — `python3 bench.py --no-cpu`", ""]
    stats = list(csv.DictReader(open(os.path.join(OUT, f"prof_{tag}", f"{tag}_kernel_stats.csv"))))
    lines += ["## --kernel-trace --stats (all kernels of the run: 2 warm-up + 5 timed steps + 3 untimed steps with one launch per variant, MRGFE_FUSED=0, for the", "per-variant figures; the per-kernel averages include the launches of the last one or two rounds, which find no busy pair and exit in ~4 us)", "",
              "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in stats:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    dom = next((r for r in stats if short(r["Name"]) == DOMINANT), None)
    summary = {"tag": tag, "dominant_kernel": DOMINANT}
    if dom:
        summary["avg_launch_ms_rocprof"] = float(dom["AverageNs"]) / 1e6
        summary["calls"] = int(dom["Calls"])
    # PMC passes (steps 1, warm-up 0): per-launch values; the full-batch launches are the largest ones
    fetch = counters(os.path.join(OUT, "pmc_fetch", "f_counter_collection.csv"))
    write = counters(os.path.join(OUT, "pmc_write", "w_counter_collection.csv"))
    sq = counters(os.path.join(OUT, "pmc_sq", "s_counter_collection.csv"))
    lines += ["", "## --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per launch; `bench.py --no-cpu --steps 1 --warmup 0`)", "",
              "| kernel | launches | FETCH_SIZE max | FETCH_SIZE mean | WRITE_SIZE max | WRITE_SIZE mean | HBM bytes per full launch = (2*FETCH + WRITE)*1024 |",
              "|---|---:|---:|---:|---:|---:|---:|"]
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE", [])
        w = write.get(k, {}).get("WRITE_SIZE", [])
        if not f and not w:
            continue
        fm, wm = (max(f) if f else 0.0), (max(w) if w else 0.0)
        lines.append(f"| `{k}` | {max(len(f), len(w))} | {fm:.1f} | {sum(f) / max(len(f), 1):.1f} | {wm:.1f} | {sum(w) / max(len(w), 1):.1f} | {(2 * fm + wm) * 1024 / 1e6:.2f} MB |")
        if k == DOMINANT:
            summary["traffic_bytes_per_full_launch"] = (2 * fm + wm) * 1024
            # same basis as bench.py's roofline.achieved (algorithmic bytes of the AVERAGE launch of a step): every step
            # issues the same launches, so the mean over the launches of the one profiled step is the per-launch traffic
            summary["traffic_bytes_per_mean_launch"] = (2 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024
            summary["fetch_size_kib_max"] = fm
            summary["write_size_kib_max"] = wm
    if sq:
        lines += ["", "## --pmc SQ counters (largest launch of each derivative kernel)", "",
                  "| kernel | SQ_WAVES | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE | VALU instr / wave | VALU busy = ACTIVE_INST_VALU*4 / (1024 SIMD * GUI_ACTIVE/8) |",
                  "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
        for k in sorted(sq):
            if "ndt_derivatives" not in k:
                continue
            v = {c: max(x) for c, x in sq[k].items()}
            waves = v.get("SQ_WAVES", 0) or 1
            gui = v.get("GRBM_GUI_ACTIVE", 0) / 8.0 or 1
            busy = v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * gui)
            lines.append(f"| `{k}` | {v.get('SQ_WAVES', 0):.0f} | {v.get('SQ_INSTS_VALU', 0):.0f} | {v.get('SQ_ACTIVE_INST_VALU', 0):.0f} | {v.get('SQ_WAVE_CYCLES', 0):.0f} | "
                         f"{v.get('SQ_BUSY_CYCLES', 0):.0f} | {v.get('GRBM_GUI_ACTIVE', 0):.0f} | {v.get('SQ_INSTS_VALU', 0) / waves:.0f} | {busy:.2f} |")
            if k == DOMINANT:
                summary["valu_instr_per_wave"] = v.get("SQ_INSTS_VALU", 0) / waves
                summary["valu_busy_estimate"] = busy
                summary["valu_busy_dominant"] = busy
    bench = os.path.join(OUT, f"bench_{tag}.json")
    if os.path.exists(bench):
        b = json.load(open(bench))
        summary["bench"] = {k: b.get(k) for k in ("value", "unit", "ms_per_step", "steps_in_flight", "value_one_step_at_a_time", "roofline", "cpu_baseline", "parity_vs_oracle",
                                                   "evaluations_per_alignment", "mean_valid_neighbours", "single_pair_latency_ms", "config3_shard", "pcl_ndt")}
        small = {}
        for bsz in (32, 64, 128):
            f = os.path.join(OUT, f"bench_{tag}_b{bsz}.json")
            if os.path.exists(f) and os.path.getsize(f):
                j = json.load(open(f))
                small[str(bsz)] = {"value": j["value"], "ms_per_step": j["ms_per_step"]}
        if small:
            summary["batch_size_sweep"] = small
            lines += ["", "## batch-size sweep (`bench.py --no-cpu --shard-steps 0 --batch B --steps 10`)", "", "| pairs per step | alignments/s | ms per step |", "|---:|---:|---:|"]
            lines += [f"| {k} | {v['value']:.0f} | {v['ms_per_step']:.3f} |" for k, v in small.items()]
            lines.append(f"| 256 | {b['value']:.0f} | {b['ms_per_step']:.3f} |")
        lines += ["", "## bench.py line of the same build (`python bench.py`)", "", "```json", json.dumps(b, indent=1), "```"]
        if dom:
            # the trace is of steps run one at a time (--in-flight 1): it is held against the HIP events of the line's own one-step-at-a-time leg, not against the
            # timed region's (two batches in flight: a launch shares the chip and its event pair spans both)
            one = b["roofline"].get("one_step_at_a_time") or b["roofline"]
            ev = one["avg_launch_ms"]
            lines += ["", f"Agreement check: HIP-event average of `{DOMINANT}` inside bench.py, steps one at a time = {ev * 1e3:.2f} us over {one.get('launches', b['roofline']['launches'])} launches; "
                          f"rocprofv3 average of the traced run (same shape) = {float(dom['AverageNs']) / 1e3:.2f} us over {dom['Calls']} launches (warm-up included); in the timed region "
                          f"(two batches in flight) the event pairs average {b['roofline']['avg_launch_ms'] * 1e3:.2f} us."]
    for kind in ("soak_filters", "soak_misc", "loop_parity_seeds", "config2_parity", "config1_parity_ranks", "ndt_fullsize_sweep"):  # profiles/soak_filters.py, soak_misc.py: the rows around the alignment against the oracle
        sf = os.path.join(OUT, f"{kind}_{tag}.json")
        if os.path.exists(sf) and open(sf).read().strip().startswith("{"):
            json.dump(json.loads(open(sf).read().strip()), open(os.path.join(ROOT, "profiles", f"{tag}_{kind}.json"), "w"), indent=1)
    extra = os.path.join(OUT, f"extra_{tag}.json")
    if os.path.exists(extra):
        ex = json.load(open(extra))
        overlap = os.path.join(OUT, f"overlap_{tag}.json")  # profiles/overlap_steps.py: several config[1] steps in flight
        if os.path.exists(overlap) and open(overlap).read().strip().startswith("{"):
            ex["config1_steps_in_flight"] = json.loads(open(overlap).read().strip())
            summary["config1_steps_in_flight"] = ex["config1_steps_in_flight"]
        json.dump(ex, open(os.path.join(ROOT, "profiles", f"{tag}_extra_measurements.json"), "w"), indent=1)
    # kernel stats of the non-headline paths (profiles/side_workloads.py): GICP (33k / 130k points), prefilter chain,
    # calc_fitness_score, loop-closure batch with getFitnessScore
    shard = os.path.join(OUT, f"prof_shard_{tag}", "s_kernel_stats.csv")
    if os.path.exists(shard):
        rows = list(csv.DictReader(open(shard)))
        lines += ["", "## BASELINE config[3] on one GPU (`rocprofv3 --kernel-trace --stats -- python3 bench.py --mode shard --steps 3 --warmup 1`), top kernels", "",
                  "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
        for r in rows[:12]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    # round 3: counter passes of one config[3] step (profiles/collect_r03.sh) — getFitnessScore kernels and the derivative kernel in shard mode
    def pmc_table(title, suffix, keep, per, per_name):
        fe = counters(os.path.join(OUT, f"pmc_fetch_{suffix}", "f_counter_collection.csv"))
        wr = counters(os.path.join(OUT, f"pmc_write_{suffix}", "w_counter_collection.csv"))
        sqc = counters(os.path.join(OUT, f"pmc_sq_{suffix}", "s_counter_collection.csv"))
        if not fe and not wr:
            return {}
        out = {}
        rows = ["", title, "", f"| kernel | launches | sum FETCH_SIZE KiB | sum WRITE_SIZE KiB | HBM bytes per {per_name} = (2*FETCH + WRITE)*1024 / {per} | VALU instr / wave | VALU busy |", "|---|---:|---:|---:|---:|---:|---:|"]
        for k in sorted(set(fe) | set(wr)):
            if not any(w in k for w in keep):
                continue
            f, w = fe.get(k, {}).get("FETCH_SIZE", []), wr.get(k, {}).get("WRITE_SIZE", [])
            v = {c: sum(x) for c, x in sqc.get(k, {}).items()}
            waves = v.get("SQ_WAVES", 0) or 1
            gui = v.get("GRBM_GUI_ACTIVE", 0) / 8.0 or 1
            busy = v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * gui)
            hbm = (2 * sum(f) + sum(w)) * 1024 / per
            out[k] = {"launches": max(len(f), len(w)), "hbm_bytes": hbm, "valu_instr_per_wave": v.get("SQ_INSTS_VALU", 0) / waves, "valu_busy": busy}
            rows.append(f"| `{k}` | {max(len(f), len(w))} | {sum(f):.0f} | {sum(w):.0f} | {hbm / 1e6:.1f} MB | {v.get('SQ_INSTS_VALU', 0) / waves:.0f} | {busy:.2f} |")
        lines.extend(rows)
        return out

    sh = pmc_table("## config[3] step, counter passes (`rocprofv3 --pmc ... -- python3 bench.py --mode shard --no-cpu --no-extras --steps 1 --warmup 1`: two steps per run; the rest of the run — "
                   "the untimed accounting steps bench.py adds — is included, see `steps_in_run`)", "shard", ("nn_fit", "ndt_derivatives", "nn_cellkey", "nn_gather", "nn_occupancy", "rs_", "ndt_leaf", "ndt_cellkey", "bbox"), 1, "run")
    if sh:
        # steps in the profiled run = launches of nn_fit_block_kernel (one fitness launch per step)
        def named(prefix):  # (the fitness kernels are templates: "nn_fit_sweep_kernel<false, 256>")
            return [v for k, v in sh.items() if k.startswith(prefix)]
        steps_in_run = max(1, sum(v["launches"] for v in named("nn_fit_block_kernel")))
        summary["shard_pmc"] = {"steps_in_run": steps_in_run, "kernels": sh}
        fit = sum(v["hbm_bytes"] for v in named("nn_fit_seed_kernel") + named("nn_fit_sweep_kernel")) / steps_in_run
        summary["fitness_sweep_traffic_bytes_per_step"] = fit
        summary["fitness_sweep_valu_busy"] = max([v["valu_busy"] for v in named("nn_fit_sweep_kernel")], default=None)
        lines += ["", f"Per step ({steps_in_run} steps in the run): `nn_fit_seed_kernel` + `nn_fit_sweep_kernel` move {fit / 1e9:.2f} GB of HBM traffic; "
                      f"`nn_fit_block_kernel` {sum(v['hbm_bytes'] for v in named('nn_fit_block_kernel')) / steps_in_run / 1e9:.2f} GB."]
    gi = pmc_table("## GICP batch, counter passes (`rocprofv3 --pmc ... -- python3 profiles/gicp_profile.py batch`: 4 aligns of 32 x ~130k-point clouds, covariances recomputed each)", "gicp",
                   ("nn_knn", "gicp_", "nn_cellkey", "nn_gather", "rs_"), 4, "align call")
    if gi:
        summary["gicp_pmc_per_align"] = gi
        if "nn_knn_kernel" in gi:
            summary["knn_traffic_bytes_per_launch"] = gi["nn_knn_kernel"]["hbm_bytes"] * 4 / max(1, gi["nn_knn_kernel"]["launches"])
            summary["knn_valu_busy"] = gi["nn_knn_kernel"]["valu_busy"]
    for name, d in (("config[3] shard of 8 (`bench.py --mode shard --shard-of 8`)", f"prof_shard8_{tag}"), ("GICP batch (`profiles/gicp_profile.py batch`)", "prof_gicp_batch"),
                    ("GICP odometry frame (`profiles/gicp_profile.py frame`)", "prof_gicp_frame")):
        f = os.path.join(OUT, d, "s_kernel_stats.csv")
        if not os.path.exists(f):
            continue
        rows = list(csv.DictReader(open(f)))
        lines += ["", f"## {name}: `rocprofv3 --kernel-trace --stats`, top kernels", "", "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
        for r in rows[:12]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    for nm in ("shard", "shard8"):
        f = os.path.join(OUT, f"bench_{nm}_{tag}.json")
        if os.path.exists(f) and os.path.getsize(f):
            j = json.load(open(f))
            summary[f"bench_{nm}"] = {"ms_per_step": j["ms_per_step"], "value": j["value"], "records_sha256_16": j["config3_shard"]["records_sha256_16"], "roofline": j["roofline"],
                                      "roofline_fitness": j["roofline_fitness"]}
            lines += ["", f"`bench.py --mode shard{' --shard-of 8' if nm == 'shard8' else ''} --no-cpu --no-extras`: {j['ms_per_step']:.2f} ms per step, records digest {j['config3_shard']['records_sha256_16']}"]
    for w in ("batch", "frame"):
        f = os.path.join(OUT, f"gicp_{w}.txt")
        if os.path.exists(f):
            summary[f"gicp_{w}"] = open(f).read().strip()
            lines += ["", f"`profiles/gicp_profile.py {w}`: {summary[f'gicp_{w}']}"]
    for w in ("gicp", "gicp_full", "prefilter", "fitness", "lc", "gicp_lc"):
        side = os.path.join(OUT, f"prof_side_{w}", "s_kernel_stats.csv")
        if not os.path.exists(side):
            continue
        rows = list(csv.DictReader(open(side)))
        lines += ["", f"## side workload `{w}` (`rocprofv3 --kernel-trace --stats -- python3 profiles/side_workloads.py {w}`), top kernels", "",
                  "| kernel | calls | total us | avg us | % |", "|---|---:|---:|---:|---:|"]
        for r in rows[:8]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    # ---- roofline from the traces alone (round 4): for each isolated workload the dominant kernel's rocprofv3 average / maximum beside the
    # HIP-event figures of the bench line the SAME profiled command printed (its log's last line)
    def last_json(path):
        if not os.path.exists(path):
            return None
        for ln in reversed(open(path).read().splitlines()):
            if ln.startswith("{"):
                try:
                    return json.loads(ln)
                except ValueError:
                    return None
        return None

    def trace_row(stats_csv, kernel):
        if not os.path.exists(stats_csv):
            return None
        for r in csv.DictReader(open(stats_csv)):
            if short(r["Name"]) == kernel:
                return r
        return None

    rft = {}
    for name, log, stats_csv in (("config1", os.path.join(OUT, f"prof_{tag}.log"), os.path.join(OUT, f"prof_{tag}", f"{tag}_kernel_stats.csv")),
                                 ("config1_two_steps_in_flight", os.path.join(OUT, f"prof_pipe_{tag}.log"), os.path.join(OUT, f"prof_pipe_{tag}", "s_kernel_stats.csv")),
                                 ("config3", os.path.join(OUT, f"prof_shard_{tag}.log"), os.path.join(OUT, f"prof_shard_{tag}", "s_kernel_stats.csv")),
                                 ("config3_shard_of_8", os.path.join(OUT, f"prof_shard8_{tag}.log"), os.path.join(OUT, f"prof_shard8_{tag}", "s_kernel_stats.csv"))):
        j, row = last_json(log), trace_row(stats_csv, DOMINANT)
        if not j or not row:
            continue
        roof = j["roofline"]
        avg_ms, max_ms = float(row["AverageNs"]) / 1e6, float(row["MaxNs"]) / 1e6
        alg = roof.get("alg_bytes_per_launch")
        rec = {"kernel": DOMINANT, "trace_calls": int(row["Calls"]), "trace_avg_launch_ms": avg_ms, "trace_max_launch_ms": max_ms, "alg_bytes_per_launch": alg,
               "frac_from_trace": (alg / 1e9) / (avg_ms / 1e3) / 8000.0 if alg else None, "frac_in_bench_line": roof.get("frac"), "hip_event_avg_launch_ms": roof.get("avg_launch_ms"),
               "hip_event_launches": roof.get("launches"), "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py " +
               {"config1": "--no-cpu --no-extras --shard-steps 0" + (" --in-flight 1" if os.path.exists(os.path.join(OUT, f"prof_pipe_{tag}.log")) else ""),
                "config1_two_steps_in_flight": "--no-cpu --no-extras --shard-steps 0 --in-flight 2 --no-seq --steps 10", "config3": "--mode shard --no-cpu --no-extras --steps 6 --warmup 2",
                "config3_shard_of_8": "--mode shard --no-cpu --no-extras --shard-of 8 --steps 12 --warmup 3"}[name]}
        if rec["frac_from_trace"] and rec["frac_in_bench_line"]:
            rec["agreement"] = rec["frac_from_trace"] / rec["frac_in_bench_line"]
        ll = roof.get("largest_launch")
        if ll:
            rec["largest_launch"] = {"alg_bytes": ll["alg_bytes"], "busy_pairs_by_kind": ll["busy_pairs_by_kind"], "hip_event_ms": ll["ms"], "frac_hip_events": ll["frac"],
                                     "frac_from_trace_max": (ll["alg_bytes"] / 1e9) / (max_ms / 1e3) / 8000.0}
        rft[name] = rec
    if rft:
        summary["roofline_from_trace"] = rft
        lines += ["", "## roofline.frac recomputed from the traces alone (isolated workloads: the dominant kernel runs only in steps of the named workload)", "",
                  "| workload | trace calls | trace avg ms | HIP-event avg ms | alg GB / launch | frac from trace | frac in the line | ratio | largest launch: trace max ms / alg GB / frac |",
                  "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
        for name, r in rft.items():
            ll = r.get("largest_launch")
            lines.append(f"| {name} | {r['trace_calls']} | {r['trace_avg_launch_ms']:.4f} | {r['hip_event_avg_launch_ms']:.4f} | {r['alg_bytes_per_launch'] / 1e9:.3f} | {r['frac_from_trace']:.3f} | "
                         f"{r['frac_in_bench_line']:.3f} | {r.get('agreement', float('nan')):.3f} | " + (f"{r['trace_max_launch_ms']:.3f} / {ll['alg_bytes'] / 1e9:.2f} / {ll['frac_from_trace_max']:.2f}" if ll else "—") + " |")
    # ---- round 5: PCL_NDT_HIP (registration_method "NDT": pcl::NormalDistributionsTransform, f64 pair terms) — trace, the workload's own line, counters
    pj = last_json(os.path.join(OUT, f"prof_pclndt_{tag}.log"))
    prow = trace_row(os.path.join(OUT, f"prof_pclndt_{tag}", "s_kernel_stats.csv"), "ndt_derivatives_f64_all_kernel")
    if pj and prow:
        avg_ms = float(prow["AverageNs"]) / 1e6
        rec = {"workload": pj, "trace_calls": int(prow["Calls"]), "trace_avg_launch_ms": avg_ms, "trace_max_launch_ms": float(prow["MaxNs"]) / 1e6,
               "frac_from_trace": (pj["alg_bytes_per_launch"] / 1e9) / (avg_ms / 1e3) / 8000.0 if pj.get("alg_bytes_per_launch") else None,
               "command": "rocprofv3 --kernel-trace --stats -- python3 profiles/pclndt_profile.py 64 1e-5 3"}
        lines += ["", "## PCL_NDT_HIP (`rocprofv3 --kernel-trace --stats -- python3 profiles/pclndt_profile.py 64 1e-5 3`: 64 config[1] pairs, eps 1e-5), top kernels", "",
                  "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
        for r in list(csv.DictReader(open(os.path.join(OUT, f"prof_pclndt_{tag}", "s_kernel_stats.csv"))))[:8]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
        lines += ["", f"`ndt_derivatives_f64_all_kernel`: trace average {avg_ms * 1e3:.1f} us over {prow['Calls']} launches (warm-up step included), HIP-event average {pj['avg_launch_ms'] * 1e3:.1f} us over "
                      f"{pj['launches']}; algorithmic bytes per launch {pj['alg_bytes_per_launch'] / 1e9:.3f} GB (N (16 + 27*8) + neighbours * 112) -> frac {rec['frac_from_trace']:.3f} from the trace, "
                      f"{pj['frac']:.3f} in the workload's own line; {pj['ms_per_step']:.2f} ms per 64-pair step, {pj['iterations_per_alignment']:.1f} iterations / {pj['evaluations_per_alignment']:.1f} evaluations per alignment."]
        pn = pmc_table("## PCL_NDT_HIP, counter passes (`rocprofv3 --pmc ... -- python3 profiles/pclndt_profile.py 64 1e-5 1`: two 64-pair steps per run)", "pclndt", ("ndt_derivatives_f64", "ndt_reduce", "ndt_plan"), 1, "run")
        if pn:
            rec["pmc"] = pn
            k = pn.get("ndt_derivatives_f64_all_kernel")
            if k and k["launches"]:
                rec["traffic_bytes_per_launch"] = k["hbm_bytes"] / k["launches"]
                rec["traffic_over_algorithmic"] = rec["traffic_bytes_per_launch"] / pj["alg_bytes_per_launch"] if pj.get("alg_bytes_per_launch") else None
        summary["pcl_ndt"] = rec
    inproc = os.path.join(OUT, f"inproc_{tag}.jsonl")
    if os.path.exists(inproc):
        rows = [json.loads(ln) for ln in open(inproc).read().splitlines() if ln.startswith("{")]
        if rows:
            summary["node_inproc"] = [{"members": r["n_gpus"], "devices": r["config"]["devices"], "ms_per_step": r["ms_per_step"], "records_sha256_16": r["records_sha256_16"],
                                       "record_gather": r["config"]["record_gather"], "blocks": r["config"]["blocks"]} for r in rows]
            lines += ["", "## mrgfe_node_* on ONE card (`python3 bench.py --mode shard --inproc --gpus N`: N members sharing device 0 — the digest is the point, not the time)", "",
                      "| members | ms per step | records digest | gather |", "|---:|---:|---|---|"]
            lines += [f"| {r['n_gpus']} | {r['ms_per_step']:.2f} | {r['records_sha256_16']} | {r['config']['record_gather']} |" for r in rows]
    soak = os.path.join(OUT, f"soak_{tag}.json")
    if os.path.exists(soak) and os.path.getsize(soak):
        sj = json.load(open(soak))
        json.dump(sj, open(os.path.join(ROOT, "profiles", f"{tag}_soak.json"), "w"), indent=1)
        a, b = sj["all_methods"], sj["pcl_gicp_and_reciprocal_icp"]
        summary["soak"] = {"ndt_over_bar": f"{a['ndt_over_bar']}/{a['ndt']}", "ndt_settled_over_bar": f"{a['ndt_settled_over_bar']}/{a['ndt_settled']}",
                           "ndt_over_bar_equal_to_gpu_order_replay": f"{a['ndt_over_bar_equal_to_gpu_order_replay']}/{a['ndt_over_bar']}",
                           "ndt_bit_identical_to_reference_order_oracle": f"{a['ndt_exact_ref']}/{a['ndt']}", "ndt_bit_identical_to_gpu_order_replay": f"{a['ndt_exact_gpu_order']}/{a['ndt']}",
                           "ndt_worst_settled": a["ndt_worst_settled"], "other_methods_over_bar": f"{a['other_over_bar']}/{a['other']}", "other_methods_bit_identical": f"{a['other_exact']}/{a['other']}",
                           "pcl_gicp_serial_over_bar": f"{b['gicp_serial_over_bar']}/{b['gicp_serial']}", "pcl_gicp_serial_bit_identical_to_reference_order_oracle": f"{b['gicp_serial_exact_ref']}/{b['gicp_serial']}",
                           "pcl_gicp_omp_over_bar": f"{b['gicp_omp_over_bar']}/{b['gicp_omp']}", "pcl_gicp_omp_over_bar_equal_to_gpu_order_replay": f"{b['gicp_omp_over_bar_equal_to_gpu_order_replay']}/{b['gicp_omp_over_bar']}",
                           "pcl_gicp_omp_bit_identical_to_gpu_order_replay": f"{b['gicp_omp_exact_gpu_order']}/{b['gicp_omp']}", "pcl_gicp_omp_worst": b["gicp_omp_worst"],
                           "pcl_gicp_flag_or_iteration_mismatch": b["gicp_flag_or_iteration_mismatch"],
                           "icp_reciprocal_over_bar": f"{b['icp_over_bar']}/{b['icp']}", "seconds": sj["seconds"]}
        if "pcl_ndt" in sj:
            c = sj["pcl_ndt"]
            summary["soak"].update({"pcl_ndt_over_bar": f"{c['over_bar']}/{c['cases']}", "pcl_ndt_bit_identical_to_reference_order_oracle": f"{c['exact']}/{c['cases']}", "pcl_ndt_worst": c["worst"],
                                    "pcl_ndt_flag_iteration_or_evaluation_mismatch": c["flag_or_iteration_mismatch"] + c["evaluation_count_mismatch"],
                                    "pcl_ndt_scenes_that_stop_after_one_iteration": f"{c['one_iteration']}/{c['cases']}"})
        if "ndt_reference_order" in sj:  # NDT_HIP with MRGFE_NDT_REFERENCE_ORDER=1 against the reference-order oracle (round 6)
            c = sj["ndt_reference_order"]
            summary["soak"].update({"ndt_reference_order_over_bar": f"{c['over_bar']}/{c['cases']}", "ndt_reference_order_bit_identical_to_reference_order_oracle": f"{c['exact']}/{c['cases']}",
                                    "ndt_reference_order_worst": c["worst"], "ndt_reference_order_flag_or_iteration_mismatch": c["flag_or_iteration_mismatch"],
                                    "ndt_reference_order_scenes_that_do_not_settle": f"{c['unsettled']}/{c['cases']}"})
        lines += ["", f"## parity soak (`python3 profiles/soak.py 3000 900 500 2000`, profiles/{tag}_soak.json)", "", "```json", json.dumps(summary["soak"], indent=1), "```"]
    for part in ("trace", "pmc"):
        f = os.path.join(OUT, f"collected_rev_{part}.txt")
        if os.path.exists(f):
            summary.setdefault("git_rev", open(f).read().strip())
            summary[f"git_rev_{part}"] = open(f).read().strip()
        f = os.path.join(OUT, f"collected_date_{part}.txt")
        if os.path.exists(f):
            summary.setdefault("collected", open(f).read().strip())
    open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
    # the raw per-kernel stats travel too (small)
    import shutil

    shutil.copy(os.path.join(OUT, f"prof_{tag}", f"{tag}_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    print(json.dumps(summary, indent=1)[:1500])


if __name__ == "__main__":
    main()
