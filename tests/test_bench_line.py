"""bench.py's stdout line is what the driver parses: it must stay small and carry the contract's keys (the 20 KB line of round 5 came
back `parsed: null`).  The line builder runs here on canned records — round 5's full record as it was printed on the GPU box
(tests/golden/bench_full_record_r05.json) and degenerate ones."""
import importlib
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


@pytest.fixture(scope="module")
def full():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r05.json")))


def strict_loads(text):
    def bad(c):
        raise ValueError(f"non-standard JSON constant {c}")
    return json.loads(text, parse_constant=bad)


def test_compact_line_fits_and_carries_the_contract(bench, full):
    assert len(json.dumps(full)) > 16000  # the record that was not parsed
    line = json.dumps(bench.compact_line(full))
    assert len(line) < bench.LINE_BUDGET == 4096 and "\n" not in line
    d = strict_loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == full["metric"] and d["unit"] == "alignments/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert abs(d["value"] - full["value"]) < 1e-4 * full["value"] and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-4 * full["ms_per_step"]
    assert d["config"]["workload"] and "model" not in d["config"] and d["config"]["steps_in_flight"] == 2 and d["config"]["pairs_per_gpu_per_step"] == 256
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "valu_busy", "avg_launch_ms", "launches", "alg_bytes_per_launch"):
        assert r[k] is not None, k
    # the kernel ALONE on the chip (the one-step-at-a-time pass), consistent with itself: achieved = bytes per launch / launch time, frac = achieved / peak
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / 1e9 / (r["avg_launch_ms"] / 1e3)) < 2e-3 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["frac"] - full["roofline"]["one_step_at_a_time"]["frac"]) < 1e-3
    assert abs(r["in_timed_region"]["frac_per_overlapped_launch"] - full["roofline"]["frac"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 64 and c["value"] > 0 and c["unit"] == "alignments/s" and c["sample"]
    assert d["parity_vs_oracle"]["pairs"] == 256 and d["parity_vs_oracle"]["pairs_over_bar"] == 0
    assert d["soak_over_bar"]["ndt"] == "0/92" and d["soak_over_bar"]["pcl_ndt"] == "0/60" and d["soak_over_bar"]["other"] == "0/68"
    assert d["value_one_step_at_a_time"] > 0 and len(d["value_host_pointers"]) == 3 and all(v > 0 for v in d["value_host_pointers"].values())
    assert d["config3"]["records_sha256_16"] == full["config3_shard"]["records_sha256_16"] and d["config3"]["pairs_over_bar"] == 0
    assert d["extras"] == bench.EXTRAS_FILE


def test_compact_line_of_degenerate_records(bench, full, monkeypatch):
    # an N-rank line: no cpu baseline, no parity, no extras
    bare = {k: full[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}
    bare.update(n_gpus=8, roofline=dict(full["roofline"], one_step_at_a_time=None), cpu_baseline=None, parity_vs_oracle=None, soak_over_bar=None)
    d = strict_loads(json.dumps(bench.compact_line(bare)))
    assert d["n_gpus"] == 8 and d["cpu_baseline"] is None and d["roofline"]["frac"] == round(full["roofline"]["frac"], 5) and "in_timed_region" not in d["roofline"]
    # NaN / inf never reach the line (strict JSON has neither)
    odd = dict(full, value=float("nan"), ms_per_step=float("inf"))
    d = strict_loads(json.dumps(bench.compact_line(odd)))
    assert d["value"] is None and d["ms_per_step"] is None
    # a record whose optional blocks are far too large sheds them instead of breaking the budget
    fat = dict(full)
    fat["config"] = dict(full["config"], workload_short="x" * 3000)
    assert len(bench.compact_line(fat)["config"]["workload"]) == 320
    monkeypatch.setattr(bench, "LINE_BUDGET", 2200)
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 2200 and "config3" not in json.loads(line) and "roofline" in json.loads(line) and "cpu_baseline" in json.loads(line)
    monkeypatch.undo()
    # the config[3] (--mode shard) line
    shard = {"metric": full["metric"], "value": 1.0, "unit": "alignments/s", "n_gpus": 2, "steps": 1, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "strong",
             "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "BASELINE config[3]"}, "roofline": full["config3_shard"]["roofline"],
             "cpu_baseline": None, "config3_shard": full["config3_shard"]}
    d = strict_loads(json.dumps(bench.compact_line(shard)))
    assert d["scaling"] == "strong" and d["config3"]["records_sha256_16"] and d["roofline"]["kernel"].startswith("ndt_derivatives")


def test_emit_prints_one_stdout_line_and_keeps_the_full_record(bench, full, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and strict_loads(lines[0])["value"] > 0
    assert err.getvalue().startswith("[bench extras] {") and json.loads(err.getvalue()[len("[bench extras] "):]) == full
    assert json.load(open(tmp_path / bench.EXTRAS_FILE)) == full
    # --full-line: the full record on stdout (tools)
    out = io.StringIO()
    with redirect_stdout(out):
        bench.emit(full, full_line=True)
    assert json.loads(out.getvalue()) == full


def test_compact_line_of_this_rounds_record_fits_with_its_newer_blocks(bench):
    """The full record of round 6's driver-shaped run (profiles/r06_bench_extras.json: with the config[2] keyframe figures and the detect() leg the line has
    grown to 3.3 KB) through the line builder: under the budget with every optional block still in it."""
    path = os.path.join(ROOT, "profiles", "r06_bench_extras.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r06_bench_extras.json is not in this tree")
    full = json.load(open(path))
    line = json.dumps(bench.compact_line(full))
    assert len(line) < bench.LINE_BUDGET
    d = strict_loads(line)
    for k in REQUIRED + ("parity_vs_oracle", "soak_over_bar", "value_host_pointers", "single_pair_latency_ms", "config3", "config2_gicp"):
        assert k in d, k
    g = d["config2_gicp"]["SMALL_GICP_HIP"]
    assert g["frame_ms_1m_from_keyframe"] < g["frame_ms"] and g["keyframe_every_metre_ms"] > 0
