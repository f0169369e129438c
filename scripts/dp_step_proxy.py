"""What a rank's data-parallel train step costs at its 512-ray share, measured on a single-rank RCCL group (development aid):
  plain      forward + loss + backward, no collective                                   (bench.py per_rank_proxy.512.*.ms_per_step)
  allreduce  the flat 2.27 MiB SUM all-reduce alone, back to back                        (bench.py allreduce_ms_single_rank)
  dp         the step NeRFRunner makes under a launcher: backward into the bucket, early part reduced on the side stream behind the
             library's event, late part after the backward (GradBucket.enable_overlap)
  dp_plain   the same with ONE collective after the backward (no overlap)
usage: python scripts/dp_step_proxy.py [rays=512] [bf16=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench
import nerf_tiny_amd as P

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bf16 = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(bench._free_port()))
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
sys.stdout.flush()
saved = os.dup(1)
os.dup2(2, 1)  # RCCL's banner
dist.init_process_group("nccl", device_id=dev)
row, col, pb, K, C_true = bench.synth_inputs(seed=1000)
model = bench.synth_weights(seed=0).to(dev)
model.batch_ray = rays
model.bf16_mlp = bf16
inp = (row[:rays].to(dev), col[:rays].to(dev), pb[:rays].float().to(dev), C_true[:rays].to(dev))


def timed(fn, n=200, w=20):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {"rays": rays, "bf16": bf16}
for name, overlap, collective in (("plain", False, False), ("dp_plain", False, True), ("dp", True, True)):
    bucket = P.parallel.GradBucket(model.network.parameters())
    if overlap:
        bucket.enable_overlap()
    model.grad_bucket = bucket

    def step():
        model.train_step(inp[0], inp[1], inp[2], K, inp[3])
        if collective:
            bucket.allreduce_sum()
        else:
            bucket.consume()

    out[name + "_ms"] = round(timed(step), 4)
    if name == "plain":
        bucket.pending = False
        out["allreduce_ms"] = round(timed(bucket.allreduce_sum), 4)
    model.grad_bucket = None
os.dup2(saved, 1)
print(out, flush=True)
dist.destroy_process_group()
