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
