#!/usr/bin/env python3
"""bench.py -- rays/s of the MI355X volume-rendering hot path on BASELINE.json's cfg2 workload.

A "step" = one full forward of the hot path (ray generation, 64 coarse + 128 fine samples per ray,
encode, 8x256 MLP per sample, resample, merge/sort, composite) over one batch of 4096 synthetic rays of
a 400x400 lego-like view, fp32, inputs already resident in HBM.  `--mode train` times forward + ray_loss +
backward instead (no optimizer), reported as an extra metric.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1: launched under torch.distributed.run,
one rank per GPU; every rank renders its own 4096-ray batches -- image-space ray batches are independent
units, so there is no data-path collective: scaling = "weak").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 1_182_976  # GEMM MACs x 2 of one MLP evaluation
B, NC, NF = 4096, 64, 128
FLOP_PER_RAY_FWD = FLOP_PER_SAMPLE * (NC + NF)  # 227,131,392
FLOP_PER_RAY_TRAIN = 676_282_368
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md chip table (fp32 matrix, dense)
PEAK_BF16_MFMA_TFLOPS = 2500.0                   # same table: bf16 matrix, dense (no sparsity)


def synth_inputs(seed):
    """cfg2: row, col ~ U{0..399}, one lego-like pose, near/far 2/6.  Pure numpy/torch; the
    same generator as oracle.lego_inputs, restated here so the product path never imports the oracle."""
    import numpy as np
    import torch

    H = W = 400
    angle = 0.6911112070083618
    focal = 0.5 * W / np.tan(0.5 * angle)
    pose = np.array([[-0.99990219, 0.00419225, -0.01334572, -0.05379832],
                     [-0.01398868, -0.29965907, 0.95394367, 3.84547043],
                     [-4.66e-10, 0.95403719, 0.29968831, 1.20808232]], dtype=np.float64)
    rng = np.random.default_rng(seed)
    row = rng.integers(0, W, size=B)
    col = rng.integers(0, H, size=B)
    m = np.concatenate((pose, np.array([[H], [W], [focal]], dtype=np.float64)), axis=1).flatten()
    pb = np.tile(np.concatenate((m, [2.0, 6.0])), (B, 1))
    K_inv = torch.tensor([[1.0, 0.0, -0.5 * W], [0.0, -1.0, 0.5 * H], [0.0, 0.0, -focal]]).float().t()
    C_true = np.random.default_rng(seed + 1).uniform(0, 1, size=(B, 3)).astype(np.float32)
    return (torch.from_numpy(row.astype(np.int64)), torch.from_numpy(col.astype(np.int64)), torch.from_numpy(pb), K_inv,
            torch.from_numpy(C_true))


def synth_weights(seed):
    """random-init weights of the reference architecture: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) per tensor."""
    import math

    import numpy as np
    import torch

    import nerf_tiny_amd as P

    m = P.NeRFModel(NC, NF, B)
    sd = m.state_dict()
    for i, (k, v) in enumerate(sd.items()):
        fan_in = v.shape[1] if v.dim() == 2 else sd[k.replace("bias", "weight")].shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        sd[k] = torch.from_numpy(np.random.default_rng([seed, i]).uniform(-bound, bound, size=tuple(v.shape)).astype(np.float32))
    m.load_state_dict(sd)
    return m


def host_cores():
    """CPU threads this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(seconds_budget=25.0):
    """The oracle (bit-identical restatement of the reference, 'port') timed on this box's host cores on the SAME
    cfg2 workload: one warm-up + as many full 4096-ray forwards as fit the budget (>= 1), best time."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    row, col, pb, K, _ = O.lego_inputs(B, seed=0)
    params = O.make_weights(0)
    with torch.no_grad():
        O.render(params, row[:256], col[:256], pb[:256], K, NC, NF)  # warm-up (small)
        best, n, t_start = None, 0, time.perf_counter()
        while n < 1 or (time.perf_counter() - t_start) < seconds_budget and n < 5:
            t0 = time.perf_counter()
            O.render(params, row, col, pb, K, NC, NF)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            n += 1
    out = {"value": round(B / best, 1), "unit": "rays/s", "cores": cores, "kind": "port",
           "sample": f"{n} full forward(s) of the same 4096-ray x (64+128) batch, torch CPU fp32, best of {n}; "
                     f"{best:.2f} s/batch"}
    if cores > 8:  # tie back to BASELINE.md section 2 (the reference itself: 967 rays/s on 8 Xeon vCPUs)
        torch.set_num_threads(8)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.render(params, row, col, pb, K, NC, NF)
            out["value_at_8_threads"] = round(B / (time.perf_counter() - t0), 1)
        torch.set_num_threads(cores)
    return out


def read_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary, if present."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(p) as f:
            return json.load(f).get("k_field_fwd", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["forward", "train"], default="forward")
    ap.add_argument("--mlp", choices=["f32", "bf16"], default="f32",
                    help="f32 = the headline metric (reference precision); bf16 = BASELINE.json cfg3 'bf16 MLP / fp32 composite' "
                         "(NOT the headline: reduced precision, reported as its own metric)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every rank renders its own 4096-ray batches; strong: ONE 4096-ray "
                         "batch is split into contiguous slices of 4096/N rays (SURVEY.md 8d cfg3), with the global ray 0's (near, far) "
                         "handed to every rank (quirk Q6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import nerf_tiny_amd as P
    from nerf_tiny_amd import _abi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    _abi.lib()  # fail loudly if the HIP library is missing
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    saved_stdout = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":  # the latter: rehearse the RCCL path with a single rank
        # RCCL prints a version banner on fd 1; the contract is ONE JSON line on stdout, so native stdout goes to stderr
        # until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # only the single-rank rehearsal comes without a launcher's environment
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    strong = args.scaling == "strong"
    row, col, pb, K, C_true = synth_inputs(seed=1000 + (0 if strong else rank))
    model = synth_weights(seed=0).to(dev)
    b_local = B
    if strong:
        if B % world:
            raise SystemExit(f"--scaling strong needs {B} % N == 0")
        b_local = B // world
        sl = slice(rank * b_local, (rank + 1) * b_local)
        ray0 = (float(pb[0, 15]), float(pb[0, 16]))  # the GLOBAL ray 0's spacing goes to every shard (nerf.py:233)
        row, col, pb, C_true = row[sl], col[sl], pb[sl], C_true[sl]
        model.batch_ray = b_local
        model.ray0_near_far = ray0
    bf16 = args.mlp == "bf16"
    model.bf16_mlp = bf16
    row, col, pb, C_true = row.to(dev), col.to(dev), pb.float().to(dev), C_true.to(dev)
    train = args.mode == "train"
    bucket = P.parallel.GradBucket(model.network.parameters()) if (train and dist is not None) else None

    def step():
        if train:
            for p in model.network.parameters():
                p.grad = None
            Cc, Cf = model(row, col, pb, K)
            loss = model.ray_loss(Cc, Cf, C_true)
            loss.backward()
            if bucket is not None:  # data-parallel trainer: one flat 2.27 MiB SUM all-reduce over RCCL/xGMI
                bucket.allreduce_sum()
        else:
            with torch.no_grad():
                model(row, col, pb, K)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    _abi.profile_begin(args.steps * 16 + 16)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = _abi.profile_end()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        rays = (B if strong else B * world) * args.steps
        value = rays / elapsed
        # dominant kernel: k_field_fwd, launched twice per step (coarse pass B*Nc samples, fine pass B*Nf samples);
        # "launch" = the average launch, so that rocprofv3's per-kernel average is directly comparable.
        ms_sum = prof.get("field_fwd_coarse", (0.0, 0))[0] + prof.get("field_fwd_fine", (0.0, 0))[0]
        n_launch = prof.get("field_fwd_coarse", (0.0, 0))[1] + prof.get("field_fwd_fine", (0.0, 0))[1]
        avg_ms = ms_sum / max(n_launch, 1)
        flop_launch = FLOP_PER_SAMPLE * b_local * (NC + NF) // 2
        achieved = flop_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        flop_ray = FLOP_PER_RAY_TRAIN if train else FLOP_PER_RAY_FWD
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "rays/sec (64 coarse + 128 fine samples), lego 400x400" + (" [train step: fwd+loss+bwd]" if train else "")
                      + (" [cfg3: bf16 MLP / fp32 composite]" if bf16 else ""),
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": ("cfg3: lego-like 400x400 view, 4096-ray batches, 64 coarse + 128 fine samples, bf16 MLP (fp32 accumulate) / fp32 "
                                    "everything else, " if bf16 else
                                    "cfg2: lego-like 400x400 view, 4096-ray batches, 64 coarse + 128 fine samples, fp32, ")
                                   + "random-init 8x256 NeRF MLP (593,924 params)", "rays_per_step_per_gpu": b_local,
                       "mode": args.mode, "parallelism": f"ray-batch x{world} (independent batches, no collective)" if not train else
                       f"ray-batch DP x{world} (one flat SUM all-reduce of 593,924 fp32 gradients per step)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None if bf16 else read_traffic(),
                         "kernel": (("k_field_fwd_bf16" if train else "k_field_fwd_bf16x") if bf16 else "k_field_fwd_reg") + " (average of the coarse- and fine-pass launches)",
                         "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                         "flop_per_launch": flop_launch},
            "whole_path_tflops": round(value / world * flop_ray / 1e12, 2),  # per GPU
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items()},
        }
        if bf16 and train and "bwd_dw" in prof:
            # the bf16 train step is HBM-bound; its dominant phase is the weight-gradient passes: 318 KiB of
            # bf16 operands per 32-sample wave block (11 passes: G and X pieces of bf16_common.h), read once each
            wb = ((b_local * NC + 255) // 256 + (b_local * NF + 255) // 256) * 8
            dw_bytes = 318 * 1024 * wb
            dw_ms = prof["bwd_dw"][0] / max(prof["bwd_dw"][1], 1)
            out["roofline"] = {"bound": "hbm", "achieved": round(dw_bytes / (dw_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(dw_bytes / (dw_ms * 1e-3) / 8e12, 4), "traffic": None,
                               "kernel": "k_dw_bf16 (the 11 weight-gradient passes + slab reduces of one step)",
                               "avg_launch_ms": round(dw_ms, 4), "launches": prof["bwd_dw"][1], "bytes_per_launch": dw_bytes}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = round(value / cb["value"], 1)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
