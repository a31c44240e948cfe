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


def test_headline_vs_reference_comparison_is_a_pure_function_of_the_two_records():
    """tools/e2e_main.vs_reference: the drop-in's epochs against profiles/r05_reference_cpu.json (the unmodified reference at the headline shape) -
    north_star's tolerances (loss 1e-4 relative, every metric +-0.002), and the dataset must be the same bytes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_main
    ref = e2e_main.reference_record()
    assert ref is not None and ref["file"].startswith("r05_") and len(ref["epochs"]) >= 2 and len(ref["digests"]) == 9
    assert ref["n_test_users"] == 13187 and ref["n_batch"] == 42 and ref["seed"] == 2022
    same = [{"loss": e["loss"], "mf_loss": e["mf_loss"], "emb_loss": 1e-7, "metrics": e["metrics"]} for e in ref["epochs"]]
    v = e2e_main.vs_reference(same, ref, ref["digests"])
    assert v["ok"] and v["loss_rel"] == 0.0 and v["metric_max_abs"] == 0.0 and v["epochs"] == len(ref["epochs"])
    # a drifted loss, a drifted metric, other bytes: each alone fails the gate
    bad = json.loads(json.dumps(same)); bad[1]["loss"] *= 1.0 + 2e-4
    assert not e2e_main.vs_reference(bad, ref, ref["digests"])["ok"]
    bad = json.loads(json.dumps(same)); bad[0]["metrics"]["recall"][1] += 0.0021
    w = e2e_main.vs_reference(bad, ref, ref["digests"])
    assert not w["ok"] and w["metric_worst"] == "epoch0/recall[1]"
    other = dict(ref["digests"], **{"train.json": "0" * 64})
    assert not e2e_main.vs_reference(same, ref, other)["ok"]


def test_topk_mode_policy():
    """ops.topk_mode: "auto" = the bf16 sweep whenever K leaves room for the verification (round 6: at every table size - tables beyond
    131 072 items are swept in item parts)."""
    from llmrec_amd import ops
    assert ops.topk_mode("auto", 17366, 64, 50) == 1 and ops.topk_mode("auto", 10322, 64, 50) == 1
    assert ops.topk_mode("auto", 524288, 64, 50) == 1 and ops.topk_mode("auto", 524289, 64, 50) == 1
    assert ops.topk_mode("auto", 1_000_000, 64, 50) == 1 and ops.topk_mode("auto", 5_000_000, 128, 50) == 1
    assert ops.topk_mode("auto", 17366, 64, 57) == 0 and ops.topk_mode("auto", 17366, 64, 56) == 1       # K <= LLMREC_TOPK_PREFILTER_MAX_K
    assert ops.topk_mode("exact", 17366, 64, 50) == 0 and ops.topk_mode("prefilter", 10**7, 64, 50) == 1
    with pytest.raises(RuntimeError):
        ops.topk_mode("fast")


def test_eight_rank_cfg4_line_is_composed_without_launching():
    """what `python bench.py --gpus 8` (cfg 4, strong scaling) will report as its shape and its exchanged bytes, from
    bench.row_sharded_plan - pure arithmetic over llmrec_amd.dist.user_block and llmrec_amd.dist_fused.plan_chunks / message_plan, the
    functions the ranks themselves use. No multi-GPU run exists; this pins the plan."""
    import bench
    p = bench.row_sharded_plan("cfg4", "strong", 8)
    c = p["config"]
    assert c["n_users_global"] == 10_000_000 and c["users_per_gpu"] == 1_250_000 and c["edges_per_gpu_nominal"] == 25_000_000
    assert c["global_batch"] == 8 * 1024 and c["user_blocks_total"] == 16
    assert [r["blocks"] for r in p["ranks"]] == [[2 * k, 2 * k + 1] for k in range(8)]
    assert [r["users"] for r in p["ranks"]] == [[1_250_000 * k, 1_250_000 * (k + 1)] for k in range(8)]          # contiguous, complete
    # one I x d message per layer and direction: 4 x 256 MB per step, each in 7 chunks (>= 32 MiB) queued behind the chunk's SpMM
    assert len(p["chunks"]) == 7 and p["chunks"][0] == (0, 142_858) and p["chunks"][-1] == (857_148, 1_000_000)
    assert all(a[1] == b[0] for a, b in zip(p["chunks"], p["chunks"][1:]))
    m = p["messages"]
    assert m["allreduce_I_x_d_bytes"] == 4 * 1_000_000 * 64 * 4 and m["allreduce_messages"] == 28 and m["chunk_bytes"] == 4 * 64 * 142_858
    assert m["bpr_rows_allgather_bytes_per_rank"] == 2 * 1024 * (256 + 8)
    # rs_ag: chunks that split evenly over the 8 ranks; the restricted forward: one message fewer, one fixed-size block more
    q = bench.row_sharded_plan("cfg4", "strong", 8, exchange="rs_ag", sparse_forward=True)
    assert all((r1 - r0) % 8 == 0 for r0, r1 in q["chunks"][:-1]) and q["messages"]["allreduce_messages"] == 3 * len(q["chunks"]) + 1
    # cfg 5 at 8 ranks, weak scaling (2 of the 16 blocks per GPU = the whole config): 2.56 GB messages, + 5 % triples
    w = bench.row_sharded_plan("cfg5", "weak", 8)
    assert w["config"]["n_users_global"] == 50_000_000 and w["config"]["augmented_triples_per_gpu"] == 51
    assert w["messages"]["allreduce_I_x_d_bytes"] == 4 * 5_000_000 * 128 * 4 and len(w["chunks"]) == 8
    # a world of one rank: nothing to overlap, one chunk
    assert len(bench.row_sharded_plan("cfg4", "strong", 1)["chunks"]) == 1
