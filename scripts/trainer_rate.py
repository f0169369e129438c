"""What the complete trainer (NeRFRunner: device sampler, forward, loss, backward, fused Adam, scheduler) sustains per batch size,
beside the device time of the same step -- the reference's own regime is BATCH_RAY = 400 (conf/lego.ini:7).
usage: python scripts/trainer_rate.py [iters]   -> one JSON line per (batch, mlp) on stdout"""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
P = importlib.import_module("nerf-tiny_amd")


def rate(batch, bf16, iters, graph):
    scene = P.data.synthetic_scene(n_pic=8, H=128, W=128)
    kw = dict(datasets={"train": scene, "val": scene, "test": scene}, batch_ray=batch, total_iter=30, step=10 ** 9, log_every=10 ** 9, bf16_mlp=bf16)
    if graph is not None:
        kw["graph_step"] = graph
    run = P.NeRFRunner(**kw)
    run.trainer("train")  # warm-up: 30 iterations
    torch.cuda.synchronize()
    run.total_iter = 30 + iters
    t0 = time.perf_counter()
    run.trainer("train")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"batch_ray": batch, "mlp": "bf16" if bf16 else "f32", "graph_step": graph, "iters": iters, "ms_per_iter": round(1e3 * dt / iters, 4),
            "rays_per_s": round(batch * iters / dt, 1)}


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    graphs = [None] if "graph_step" not in P.NeRFRunner.__init__.__code__.co_varnames else [False, True]
    for batch in (400, 512, 4096):
        for bf16 in (False, True):
            for g in graphs:
                print(json.dumps(rate(batch, bf16, iters if batch < 4096 else max(iters // 4, 20), g)), flush=True)
