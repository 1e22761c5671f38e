cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 200 python bench.py --mode shard --steps 6 --warmup 2 "$@" > gpurun_out/r3_e2_$name.json 2> gpurun_out/r3_e2_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r3_e2_$name.json"))["config3_shard"]
f=d["fitness_passes_last_step"]
print("$name", round(d["ms_per_step"],2), d["per_step_ms"], d["records_sha256_16"], "block %.2f sweep %.2f far %.2f queued %d"%(f["ms_block"],f["ms_sweep"],f["ms_far"],f["queued"]))
PY
}
run g1
run g2 --shard-of 2
run g4 --shard-of 4
run g8 --shard-of 8
