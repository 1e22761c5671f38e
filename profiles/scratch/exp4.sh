cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fitness_passes.py tests/test_gpu_batch.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3_s9_pytest.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r3_s9_pytest.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --mode shard --steps 6 --warmup 2 $EXTRA > gpurun_out/r3_e4_$name.json 2> gpurun_out/r3_e4_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_e4_$name.json"))["config3_shard"]
    f=d["fitness_passes_last_step"]
    print("$name", round(d["ms_per_step"],2), d["per_step_ms"], d["records_sha256_16"], "launches %d block %.2f sweep %.2f far %.2f"%(f["launches"],f["ms_block"],f["ms_sweep"],f["ms_far"]))
except Exception as e: print("$name failed", e)
PY
}
run early MRGFE_FIT_STATS=2
run noearly MRGFE_NO_EARLY_FIT=1
EXTRA="--shard-of 8"
run g8early A=1
run g8noearly MRGFE_NO_EARLY_FIT=1
EXTRA="--shard-of 4"
run g4early A=1
EXTRA="--shard-of 2"
run g2early A=1
