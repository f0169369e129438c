"""GPU: the RCCL leg of the multi-GPU path, run for real on the one GPU of the box in FRESH child processes (a process group per
child; never re-exec'd from a process that has initialised the GPU).  The N-rank launcher itself is covered on CPU by
tests/test_bench_launch.py; sharding arithmetic by tests/test_parallel_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.timeout(600)
def test_train_step_sharded_over_rccl_single_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_single_rank.py")], capture_output=True, text=True,
                       env=_env(), timeout=560)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL-OK" in r.stdout


@pytest.mark.timeout(900)
def test_bench_train_leg_with_rccl_allreduce():
    """bench.py's data-parallel train leg with a real (1-rank) RCCL group: one JSON line, n_gpus from the process group, the
    all-reduce measured, and every extra leg present with its own roofline block."""
    env = _env()
    env["BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=860)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["mode"] == "train"
    assert out["allreduce_ms"] is not None and 0.0 < out["allreduce_ms"] < 50.0
    assert set(out["roofline_phases"]) == {"forward_with_saves", "dx_chain", "dw"}
    # the weak legs and -- the code path of the driver's N = 8 line -- the strong-scaling form of all four legs (here: one rank = the whole batch)
    assert set(out["extra"]) == {"forward_f32", "forward_bf16", "train_bf16", "forward_f32_split", "strong_forward_f32", "strong_train_f32", "strong_forward_bf16",
                                 "strong_train_bf16", "strong_forward_f32_split"}
    for name, leg in out["extra"].items():
        if name.startswith("strong_"):
            assert leg["value"] > 0 and leg["rays_per_step"] == 4096 and 0.0 < leg["roofline_frac"] < 1.125  # (algorithmic FLOPs: 9/8 of the executed ones)
        else:
            assert leg["value"] > 0 and 0.0 < leg["roofline"]["frac"] < 1.125 and leg["roofline"].get("frac_executed", 0.0) < 1.0
    assert out["extra"]["train_bf16"]["allreduce_ms"] is not None
    assert out["extra"]["strong_train_f32"]["allreduce_ms"] is not None and out["extra"]["strong_train_bf16"]["allreduce_ms"] is not None
    assert out["parity"]["pass"] is True and out["parity"]["split_mlp_vs_reference"]["pass"] is True
