"""CPU: `bench.py --gpus N` really starts N ranks (the driver's literal command line), rehearsed over gloo with no kernels."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(300)
def test_gpus_flag_launches_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout
    assert len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["flag_gpus"] == 2 and out["steps"] == 3 and out["dry"] is True


@pytest.mark.timeout(300)
def test_launcher_reports_a_failing_rank():
    """a rank that dies makes the parent exit non-zero (here: WORLD_SIZE disagrees with --gpus inside the children)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_TEST_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode != 0


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_train_traffic_lookup_matches_the_shipped_kernel_names():
    """bench.py reads the weight-gradient phase's HBM bytes from the committed PMC summary by kernel name: a renamed kernel
    (or a summary collected before the rename) would silently turn `roofline.traffic` into null."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f32_train, f32_fwd, bf_train, bf_fwd = bench.Leg("t", True, False), bench.Leg("f", False, False), bench.Leg("bt", True, True), bench.Leg("bf", False, True)
    t = bench.read_traffic(f32_train, list(bench.DW_LAUNCHES), scale=bench.DW_LAUNCHES)
    assert t is not None and 10e9 < t < 40e9, t  # 16.2 GB per step measured (profiles/r03_train_f32_pmc.json)
    for keys in (["k_field_fwd_reg<true, false>"], ["k_field_bwd_reg<true>", "k_field_bwd_reg<false>"]):
        assert bench.read_traffic(f32_train, keys) is not None, keys
    assert bench.read_traffic(f32_fwd, ["k_field_fwd"]) is not None
    # the bf16-MLP legs (cfg3) carry measured traffic as well: no null, no computed stand-in
    tb = bench.read_traffic(bf_train, list(bench.DW_BF16_LAUNCHES), scale=bench.DW_BF16_LAUNCHES)
    assert tb is not None and 4e9 < tb < 12e9, tb  # 7.4 GB per step measured (profiles/r04_train_bf16_pmc.json)
    for keys in (["k_field_fwd_bf16<true, 8>"], ["k_field_bwd_bf16<true, 8>", "k_field_bwd_bf16<false, 8>"]):
        assert bench.read_traffic(bf_train, keys) is not None, keys
    assert bench.read_traffic(bf_fwd, ["k_field_fwd_bf16x<2, 8>"]) is not None
    # the split-fp32 train step's blocks (profiles/r05_train_split_pmc.json)
    sp_train = bench.Leg("st", True, False, True)
    ts = bench.read_traffic(sp_train, list(bench.DW_SPLIT_LAUNCHES), scale=bench.DW_SPLIT_LAUNCHES)
    assert ts is not None and 10e9 < ts < 20e9, ts  # 14.6 GB per step measured
    for keys in (["k_field_fwd_split<true>"], ["k_field_bwd_split<true>", "k_field_bwd_split<false>"]):
        assert bench.read_traffic(sp_train, keys) is not None, keys


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")


def test_result_line_stays_under_4_kb_on_a_full_record():
    """BENCH_r04.json: the driver did not take a 21 KB line.  compact_line() of a REAL full record (round 4's N = 1 default run, committed as
    profiles/r04_bench_default.json) and of its N = 8 form (five strong_* legs added, long free-text values) must stay below 4 KB and keep
    the contract's keys, `roofline` and `cpu_baseline` with the fields the contract names."""
    bench = _bench_module()
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000  # the record that broke the driver's parse
    line = bench.compact_line(full, "/somewhere/bench_extra.json")
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) < 4096, len(s)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert set(line["config"]) == {"workload", "rays_per_step_per_gpu", "mode", "weights", "parallelism"} and "model" not in line["config"]
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"] <= 1.0
    assert line["parity"]["pass"] is True and line["side_file"] == "bench_extra.json"
    assert line["train_f32_rays_per_s"] == full["extra"]["train_f32"]["value"]
    assert line["implied_strong_scaling_8_min"] == min(full["implied_strong_scaling_8"].values())
    # the N = 8 shape of the record: strong_* legs beside the weak ones, an all-reduce, pathological free text
    big = json.loads(json.dumps(full))
    for name in ("forward_f32", "train_f32", "forward_bf16", "train_bf16", "forward_f32_split"):
        src = big["extra"].get(name, big["extra"]["train_f32"])
        big["extra"]["strong_" + name] = dict(src, allreduce_ms=0.0612)
    big["n_gpus"], big["allreduce_ms"] = 8, 0.0588
    big["config"]["workload"] = "w" * 5000
    big["roofline"]["kernel"] = "k" * 5000
    big["cpu_baseline"]["sample"] = "s" * 5000
    big["metric"] = "m" * 5000
    s8 = json.dumps(bench.compact_line(big, None), separators=(",", ":"))
    assert len(s8) < 4096, len(s8)
    l8 = json.loads(s8)
    assert l8["strong_train_bf16_allreduce_ms"] == 0.0612 and l8["allreduce_ms"] == 0.0588 and l8["n_gpus"] == 8


def test_side_file_holds_the_full_record(tmp_path):
    bench = _bench_module()
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as f:
        full = json.load(f)
    p = bench.write_side_file(full, str(tmp_path / "x.json"))
    with open(p) as f:
        assert json.load(f) == full
