"""Ray-batch data parallelism over the GPUs of one node: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

Rays are independent units, so
  * inference shards image-space rays contiguously over ranks with NO collective in the data path (each rank writes
    its own rows of the frame; `gather_rows` is only for assembling the picture afterwards);
  * training splits each B-ray batch into `world` equal contiguous slices; every rank runs forward + backward on its
    slice and the 24 gradient tensors (593,924 fp32 = 2.27 MiB) are SUM-all-reduced in one flat bucket.  The loss is
    a SUM over rays (nerf.py:328-331), so SUM -- not torch DDP's mean -- reproduces the single-GPU gradient.
The only cross-ray term of the reference is the ray-0 spacing in the resampler (quirk Q6, nerf.py:233): the global
ray 0's (near, far) is handed to every shard (`NeRFModel.ray0_near_far`).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of n units for `rank`; the first n % world ranks get one extra unit."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_ray0(poses_bound: torch.Tensor) -> tuple[float, float]:
    """(near, far) of ray 0 of the GLOBAL batch in the precision the kernels see (fp32, nerf.py:338)."""
    pb0 = poses_bound[0].to(torch.float32)
    return float(pb0[15]), float(pb0[16])


class GradBucket:
    """One flat fp32 buffer aliasing the gradients of `params` (in parameters() order) so that a single
    all-reduce moves all 2.27 MiB (latency-bound on xGMI: one collective instead of 24)."""

    def __init__(self, params):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def pack(self):
        for v, p in zip(self.views, self.params):
            v.copy_(p.grad)

    def unpack(self):
        for v, p in zip(self.views, self.params):
            p.grad.copy_(v)

    def allreduce_sum(self, group=None):
        self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.unpack()


def train_step_sharded(model, bucket: GradBucket, row, col, poses_bound, K_inv, C_true, rank: int, world: int, group=None):
    """One data-parallel train step on this rank's slice of a GLOBAL batch (every rank passes the same global tensors).
    `model.batch_ray` must equal the slice size.  Returns this rank's (C_coarse, C_fine, local loss); after the call
    every rank holds the full-batch gradient in p.grad (sum over ranks)."""
    lo, hi = shard_bounds(row.shape[0], rank, world)
    model.ray0_near_far = global_ray0(poses_bound)
    for p in bucket.params:
        p.grad = None
    C_c, C_f = model(row[lo:hi], col[lo:hi], poses_bound[lo:hi], K_inv)
    loss = model.ray_loss(C_c, C_f, C_true[lo:hi])
    loss.backward()
    if world > 1:
        bucket.allreduce_sum(group)
    return C_c, C_f, loss


def render_rows_sharded(model, row, col, poses_bound, K_inv, rank: int, world: int, out: torch.Tensor | None = None):
    """Inference over a long list of rays (e.g. one frame): this rank renders the contiguous slice
    [lo, hi) in batches of model.batch_ray (the tail batch is padded by repeating the last ray and cropped -- the
    reference would silently drop it, nerf.py:442).  No collective.  Returns (lo, hi, C_fine[hi-lo, 3])."""
    lo, hi = shard_bounds(row.shape[0], rank, world)
    Bm = model.batch_ray
    model.ray0_near_far = global_ray0(poses_bound)
    res = []
    with torch.no_grad():
        for s in range(lo, hi, Bm):
            e = min(s + Bm, hi)
            idx = torch.arange(s, s + Bm).clamp_max(e - 1)
            _, C_f = model(row[idx], col[idx], poses_bound[idx], K_inv)
            res.append(C_f[: e - s].clone())
    C = torch.cat(res) if res else torch.empty(0, 3)
    if out is not None:
        out[lo:hi] = C
    return lo, hi, C


def gather_rows(C_local: torch.Tensor, n_total: int, rank: int, world: int, group=None) -> torch.Tensor:
    """Assemble the full [n_total, 3] picture on every rank (outside the timed data path)."""
    if world == 1:
        return C_local
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros(mx, 3, dtype=C_local.dtype, device=C_local.device)
    pad[: C_local.shape[0]] = C_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)])
