# Where the wavefronts of a config[3] step spend their cycles, per kernel: issuing / waiting to issue / parked on memory or a barrier
# (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES; MI355X_MICROARCH.md: the three are disjoint shares of a wavefront's cycles)
#   gpurun -- "bash profiles/wait_split.sh"      -> gpurun_out/wait_split.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --mode shard --prepare-only > /dev/null 2>&1
MRGFE_NO_EARLY_FIT=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_wait -o p -- python3 bench.py --mode shard --no-cpu --no-extras --steps 1 --warmup 1 > gpurun_out/pmc_wait.log 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/wait_split.md
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc_wait/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("mrgfe::", "").replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
print("# config[3] step (256 loop-closure pairs, fitness(inf), MRGFE_NO_EARLY_FIT=1): shares of the wavefronts' cycles per kernel (profiles/wait_split.sh; counters summed over the run's launches)\n")
print("| kernel | launches | wave cycles (quad-cycles, all launches) | issuing | waiting to issue | parked (memory / barrier) | of issuing: VALU / scalar / LDS |")
print("|---|---:|---:|---:|---:|---:|---|")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:16]:
    t = agg[k]; wc = t.get("SQ_WAVE_CYCLES", 0) or 1; a = t.get("SQ_ACTIVE_INST_ANY", 0) or 1
    print(f"| `{k}` | {cnt[k]} | {wc:.3g} | {t.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {t.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {t.get('SQ_WAIT_ANY', 0) / wc:.2f} | {t.get('SQ_ACTIVE_INST_VALU', 0) / a:.2f} / {t.get('SQ_ACTIVE_INST_SCA', 0) / a:.2f} / {t.get('SQ_ACTIVE_INST_LDS', 0) / a:.2f} |")
PY
cat gpurun_out/wait_split.md
