"""The bench's ONE stdout line stays small and strict (a 20.5 KB line came back from the driver unparsed).
bench.compact_line is a pure function of the detail record; the canned records are full lines of earlier rounds (profiles/)."""
import json
import math
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = ["r04_bench_driver_cmd.json", "r04_bench_nf_final.json", "r04_bench_ml.json", "r04_bench_cfg4_full_1gpu.json", "r04_bench_cfg5_full_1gpu.json"]


def _load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        text = f.read().strip()
    return json.loads(text.splitlines()[-1]) if not text.startswith("{\n") else json.loads(text)


def _no_nonfinite(text):
    for tok in ("NaN", "Infinity", "-Infinity"):
        assert tok not in text


@pytest.mark.parametrize("name", CANNED)
def test_compact_line_is_small_strict_json(name):
    full = _load(name)
    assert len(json.dumps(full)) > bench.COMPACT_LIMIT or "row_sharded" not in full     # (the canned record really is the long form)
    out = bench.compact_line(full)
    text = json.dumps(out, allow_nan=False)
    assert len(text) < bench.COMPACT_LIMIT, len(text)
    back = json.loads(text)
    assert back == out
    _no_nonfinite(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config"):
        assert k in back, k
    assert back["config"]["workload"] == full["config"]["workload"]
    assert all(not isinstance(v, (dict, list)) for v in back["config"].values())
    assert abs(back["value"] - full["value"]) <= 1e-6 * abs(full["value"])
    if "roofline" in full:
        r = back["roofline"]
        for k in ("kernel", "bound", "achieved", "peak", "unit", "frac"):
            assert k in r, k
        assert len(r["kernel"]) <= 48
    if "cpu_baseline" in full:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k


def test_compact_line_carries_the_round5_blocks_and_drops_nonfinite():
    full = _load("r04_bench_driver_cmd.json")
    full["roofline"]["frac"] = float("nan")
    full["roofline"]["in_step_us"] = 133.7
    full["roofline"]["frac_isolated"] = 0.42
    full["eval"]["ms"] = float("inf")
    full["propagated_edges_per_sec"] = {"executed": 1.9e9, "executed_per_step": 9.1e5, "reference_equivalent": 4.6e9, "reference_equivalent_per_step": 2205840,
                                        "definition": "x" * 500}
    full["end_to_end"]["default"]["vs_reference"] = {"ok": True, "epochs": 2, "loss_rel": 3e-6, "metric_max_abs": 0.0, "recall20": [0.1, 0.2], "note": "y" * 900}
    full["step_in_graph"] = {"entry_point_calls": 31, "projection_us": 135.2, "wgrad_us": 118.0, "how": "z" * 300}
    full["row_sharded"]["strong"]["vs_prev"] = "r03 51.0 ms was the row-restricted forward"
    out = bench.compact_line(full)
    text = json.dumps(out, allow_nan=False)
    assert len(text) < bench.COMPACT_LIMIT
    assert out["roofline"]["frac"] is None and out["eval"]["ms"] is None
    assert out["roofline"]["in_step_us"] == 133.7 and out["roofline"]["in_step_us_rocprof"] == 133.66
    assert out["propagated_edges_per_sec"] == {"executed": 1.9e9, "executed_per_step": 9.1e5, "reference_equivalent": 4.6e9, "reference_equivalent_per_step": 2205840}
    assert out["end_to_end"]["default"]["vs_reference"] == {"ok": True, "epochs": 2, "loss_rel": 3e-6, "metric_max_abs": 0.0}
    assert out["step_in_graph"]["entry_point_calls"] == 31
    assert out["row_sharded"]["strong"]["vs_prev"].startswith("r03")
    assert "definition" not in text and "note" not in text


def test_compact_line_sheds_blocks_rather_than_overflow():
    full = _load("r04_bench_driver_cmd.json")
    full["config"].update({"k%d" % i: i for i in range(400)})             # an absurd config: the guard still bounds the line
    full["cpu_baseline"]["sample"] = "s" * 5000
    out = bench.compact_line(full)
    assert len(json.dumps(out, allow_nan=False)) < bench.COMPACT_LIMIT
    assert len(out["config"]) <= 14


def test_round_helper():
    assert bench._r(float("nan")) is None and bench._r(float("-inf")) is None
    assert bench._r(True) is True and bench._r(3) == 3 and bench._r("x") == "x" and bench._r(None) is None
    assert bench._r(0.123456789, 4) == 0.1235 and bench._r(2139167.692349985, 7) == 2139168.0
    assert math.isclose(bench._r(5.681170173345285e-07, 3), 5.68e-07)
