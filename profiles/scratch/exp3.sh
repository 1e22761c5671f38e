cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --mode shard --steps 6 --warmup 2 $EXTRA > gpurun_out/r3_e3_$name.json 2> gpurun_out/r3_e3_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_e3_$name.json"))["config3_shard"]
    print("$name", round(d["ms_per_step"],2), d["per_step_ms"], d["records_sha256_16"])
except Exception as e: print("$name failed", e)
PY
}
run s1 BENCH_SUBBATCHES=1
run s2 BENCH_SUBBATCHES=2
run s4 BENCH_SUBBATCHES=4
run s4st BENCH_SUBBATCHES=4 BENCH_STAGGER_MS=2
run s8 BENCH_SUBBATCHES=8
EXTRA="--shard-of 8"
run g8s1 BENCH_SUBBATCHES=1
run g8s2 BENCH_SUBBATCHES=2
run g8s4 BENCH_SUBBATCHES=4
