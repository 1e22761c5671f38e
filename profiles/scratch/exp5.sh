cd $GRAFT_REPO_ROOT
grep -m1 "model name" /proc/cpuinfo
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --mode shard --steps 4 --warmup 1 $EXTRA > gpurun_out/r3_e5_$name.json 2> gpurun_out/r3_e5_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_e5_$name.json"))["config3_shard"]
    print("$name", round(d["ms_per_step"],2), d["records_sha256_16"], d["inputs_sha256_16"], d["mean_iterations"])
except Exception as e: print("$name failed", e)
PY
}
run noearly MRGFE_NO_EARLY_FIT=1
run early A=1
run oldfit MRGFE_FIT_SWEEP=0 MRGFE_NO_EARLY_FIT=1
