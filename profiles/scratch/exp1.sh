cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --mode shard --steps 4 --warmup 1 > gpurun_out/r3_e1_$name.json 2> gpurun_out/r3_e1_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r3_e1_$name.json"))["config3_shard"]
f=d["fitness_passes_last_step"]
print("$name", round(d["ms_per_step"],2), d["records_sha256_16"], "block %.2f sweep %.2f far %.2f queued %d"%(f["ms_block"],f["ms_sweep"],f["ms_far"],f["queued"]))
PY
}
run base A=1
run b8 MRGFE_FIT_BUILDERS=8
run c0125 MRGFE_NN_CELL=0.125
run c025 MRGFE_NN_CELL=0.25
run c025b8 MRGFE_NN_CELL=0.25 MRGFE_FIT_BUILDERS=8
run c05 MRGFE_NN_CELL=0.5
