# SQ counters of the target-build kernels (one build of 256 targets after two warm-ups: profiles/build_profile.py 256 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_build1 -o p -- python3 profiles/build_profile.py 256 1 > gpurun_out/pmc_build1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_build2 -o p -- python3 profiles/build_profile.py 256 1 > gpurun_out/pmc_build2.log 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/build_pmc.md
import csv, collections
print("# SQ counters of the NDT target-build kernels (profiles/build_pmc.sh: `profiles/build_profile.py 256 1`, three builds of 256 targets per pass; values per launch, summed over the counter instances rocprofv3 reports)\n")
tabs = {}
for d in ("pmc_build1", "pmc_build2"):
    rows = list(csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("mrgfe::", "").replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for k in agg:
        tabs.setdefault(k, {}).update({c: v / cnt[k] for c, v in agg[k].items()})
keep = [k for k in tabs if k.startswith(("ndt_", "rs_", "bbox_", "scan_"))]
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]
print("| kernel | " + " | ".join(cols) + " | issuing / waiting to issue / parked (share of wave cycles) |")
print("|---|" + "---:|" * (len(cols) + 1))
for k in sorted(keep, key=lambda k: -tabs[k].get("SQ_WAVE_CYCLES", 0)):
    t = tabs[k]; wc = t.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"| `{k}` | " + " | ".join(f"{t.get(c, 0):.0f}" for c in cols) + f" | {t.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} / {t.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {t.get('SQ_WAIT_ANY', 0) / wc:.2f} |")
PY
cat gpurun_out/build_pmc.md
