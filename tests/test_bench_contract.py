"""The committed bench line (profiles/r2_bench_n1.json, written by `python bench.py` on the GPU box) keeps the
contract the driver parses: metric / unit of BASELINE.json, whole-job value, and the `roofline` and `cpu_baseline`
objects.  Runs on CPU: it checks the artifact, not the GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r2_bench_n1.json")
if not os.path.exists(LINE):
    LINE = os.path.join(ROOT, "profiles", "r1_bench_n1.json")


@pytest.fixture(scope="module")
def line():
    if not os.path.exists(LINE):
        pytest.skip("no committed bench line")
    with open(LINE) as f:
        rows = [r for r in f.read().strip().split("\n") if r.startswith("{")]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    return json.loads(rows[0])


def test_top_level_fields(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert base["metric"].startswith(line["metric"])          # BASELINE.json appends the configuration to the metric name
    assert line["unit"] == "QA-pairs/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None                           # BASELINE.md holds no published number for this metric
    assert line["data"] == "synthetic" and line["dtype"] == "bf16"
    assert "workload" in line["config"] and "model" not in line["config"]
    # value = QA pairs of all ranks / max-over-ranks step time
    per_step = line["config"]["global_batch"]
    assert abs(line["value"] - per_step / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def test_roofline_object(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] == 2500.0 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic FLOPs per launch / measured launch duration
    assert abs(r["achieved"] - r["gflop_per_launch"] / r["us_per_launch"] * 1e3) / r["achieved"] < 1e-6    # GFLOP / us = PFLOP/s


def test_cpu_baseline_object(line):
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == line["unit"]
    assert c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str) and c["sample"]
