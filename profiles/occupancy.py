"""Average resident wavefronts and VALU / LDS busy fraction per kernel from a rocprofv3 --pmc run that collected
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE (+ --kernel-trace):   python3 profiles/occupancy.py <dir>
avg waves = SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE / 8 XCDs)  (of 8192 slots); busy = ACTIVE_INST * 4 / (1024 SIMDs * GUI cycles)."""
import csv, glob, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
d, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mrgfe::", "")[:44]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
print(f"{'kernel':44s} {'calls':>5s} {'Mcycles':>8s} {'avg waves':>9s} {'VALU':>5s} {'LDS':>5s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:24]:
    gui = v["GRBM_GUI_ACTIVE"] / 8
    if gui > 0:
        print(f"{k:44s} {len(n[k]):5d} {gui / 1e6:8.2f} {v['SQ_WAVE_CYCLES'] * 4 / gui:9.0f} {v['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * gui):5.2f} {v['SQ_ACTIVE_INST_LDS'] * 4 / (1024 * gui):5.2f}")
