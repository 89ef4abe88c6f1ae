"""bench.py's last stdout line is what the driver parses, out of the last 8 000 characters it keeps: the line is built from the full record
of a run by bench.compact_line() and has to stay below bench.LINE_LIMIT whatever the record carries (round 5's 28 KB line was recorded as
`parsed: null`).  The recorded full record is round 5's own (profiles/r13w_bench.json)."""
import contextlib
import copy
import io
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "r13w_bench.json")

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "passes", "roofline", "cpu_baseline", "north_star_target")
ROOFLINE = ("kernel", "avg_launch_ms", "bound", "achieved", "peak", "unit", "frac", "frac_vs_measured_mix", "traffic", "hbm", "algorithmic_bytes_per_launch", "int", "pmc_tree")


def _strings(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)
    elif isinstance(o, str):
        yield o


def test_line_from_a_recorded_run_is_short_and_complete():
    full = json.load(open(RECORD))
    assert len(json.dumps(full)) > 20000                      # the record that defeated the driver
    line = bench.compact_line(full, "profiles/bench_last_full.json")
    txt = json.dumps(line)
    assert len(txt) < 6000 and len(txt) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype"):
        assert line[k] == full[k]
    assert line["config"]["workload"].startswith("configs[1]")
    assert "model" not in line["config"]
    r = line["roofline"]
    for k in ROOFLINE:
        assert k in r, k
    assert r["frac"] == full["roofline"]["frac"] and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert abs(r["hbm"]["frac"] - full["roofline"]["hbm"]["frac"]) < 1e-4
    assert r["int"]["sketch_kernel_frac"] == full["roofline"]["int"]["sketch_kernel_frac"]
    assert r["pmc_tree"]["unchanged"] is True
    assert {k["kernel"] for k in r["kernels"]} == {"k_lookup_l1", "k_l2_locate", "k_l2_sweep"}
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 16 and c["threads"] == 8 and 0.1 < c["value"] < 0.3 and len(c["sample"]) <= 100
    ns = line["north_star_target"]
    assert abs(ns["value"] - 115.8828) < 1e-3 and ns["seg10000"]["value"] > 150 and ns["repeat_rich"]["hbm_point_path_share"] > 0.3
    assert ns["cpu"]["kind"] == "reference"
    assert max(len(s) for s in _strings(line)) <= 200         # numbers, not essays


def test_line_stays_short_when_the_record_grows():
    full = json.load(open(RECORD))
    full["configs2"] = copy.deepcopy(full["north_star_target"]["segLength_10000"])
    for side in (full, full["north_star_target"], full["north_star_target"]["repeat_rich"], full["configs2"]):
        side["kernels"].update({"extra_kernel_%d" % i: {"ms_per_step": 1.0 + i, "launches_per_step": 1.0} for i in range(40)})
        side["roofline"]["kernels"] = side["roofline"]["kernels"] * 6
    full["config"]["workload"] = full["config"]["workload"] * 20
    full["cpu_baseline"]["sample"] = full["cpu_baseline"]["sample"] * 20
    line = bench.compact_line(full, "profiles/bench_last_full.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line, k
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] > 0


def test_line_survives_failed_side_measurements():
    full = json.load(open(RECORD))
    full["north_star_target"] = {"error": "child exited with 1"}
    full["e2e"] = {"error": "x" * 5000}
    full["cpu_baseline"] = {"error": "y" * 5000}
    del full["host_path"]
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT and line["value"] == full["value"] and line["roofline"]["kernel"] == "k_sketch_fast"
    assert "error" in line["cpu_baseline"] and "error" in line["e2e"]


def test_emit_prints_the_parsed_line_last(tmp_path, monkeypatch):
    full = json.load(open(RECORD))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.emit(full)
    out = buf.getvalue()
    lines = out.strip().split("\n")
    last = json.loads(lines[-1])
    assert last["metric"] == full["metric"] and "roofline" in last and "cpu_baseline" in last and len(lines[-1]) < 6000
    assert all("side" in json.loads(l) for l in lines[:-1])                  # one small object per side measurement, before the line
    assert json.loads(out[-8000:].strip().split("\n")[-1]) == last           # what survives the driver's 8 000-character tail
    assert json.load(open(tmp_path / "profiles" / "bench_last_full.json")) == full
    assert last["full"] == "profiles/bench_last_full.json"


def test_line_from_the_round_6_record_carries_configs2():
    """the full record of round 6's final default run (profiles/r14_18_bench_full.json: configs[1] + e2e + north_star variants + configs2)"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r14_18_bench_full.json")))
    line = bench.compact_line(full, "profiles/bench_last_full.json")
    assert len(json.dumps(line)) < 6000
    for k in CONTRACT + ("configs2", "e2e", "host_path"):
        assert k in line, k
    c2 = line["configs2"]
    assert c2["value"] > 150 and c2["e2e"]["paf_lines"] > 0 and c2["roofline"]["pmc_unchanged"] is True
    assert line["roofline"]["pmc_tree"]["unchanged"] is True and line["roofline"]["bound"] == "valu"
    assert line["north_star_target"]["repeat_rich"]["hbm_point_path_share"] < 0.08
    assert line["passes"]["redone"] == 0 and line["config"]["parallelism"] == "single GPU"
