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

import weakref

import torch
import torch.distributed as dist

_LIVE_BUCKETS = weakref.WeakSet()  # every GradBucket alive in this process (consume_buckets_of)


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of n units for `rank`; the first n % world ranks get one extra unit."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_ray0(poses_bound: torch.Tensor) -> tuple[float, float]:
    """(near, far) of ray 0 of the GLOBAL batch in the precision the kernels see (fp32, nerf.py:338)."""
    pb0 = poses_bound[0].to(torch.float32)
    return float(pb0[15]), float(pb0[16])


EARLY_TENSORS = 16  # point_layer[0..7].{weight, bias}: final before the rest (include/nerf_hip.h: nerf_hip_backward_overlap)


class GradBucket:
    """One flat fp32 buffer holding the gradients of `params` (in parameters() order) so that a single all-reduce moves
    all 2.27 MiB (latency-bound on xGMI: one collective instead of 24).  With ``model.grad_bucket = bucket`` the backward
    kernels write straight into `views` and ``p.grad`` ARE those views, so `allreduce_sum` is the collective alone;
    gradients that live elsewhere (plain autograd, the gloo tests) are copied in and out."""

    ALIGN = 64  # floats: every view starts on a 256-byte boundary (the C ABI wants dweights24 16-byte aligned; sigma bias has 1 element)

    def __init__(self, params):
        self.params = list(params)
        p0 = self.params[0]
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.numel = sum(p.numel() for p in self.params)  # 593,924 real gradients; the padding stays zero
        self.flat = torch.zeros(o, dtype=p0.dtype, device=p0.device)
        self.views = [self.flat[a:a + p.numel()].view_as(p) for a, p in zip(offs, self.params)]
        #: a backward has written gradients into the views that nothing has consumed yet (all-reduce / optimizer step).  A second
        #: backward into the same bucket would OVERWRITE them (the kernels store, they do not accumulate): _RenderFn.backward raises.
        self.pending = False
        #: first float of the LATE part of the flat buffer: tensors 0..15 (point_layer[0..7], 83 % of the parameters) are final
        #: before the last weight-gradient products of a backward run (nerf_hip_backward_overlap), tensors 16.. after it
        self.early_numel = offs[EARLY_TENSORS] if len(offs) > EARLY_TENSORS else o
        self.early_event = None   # torch.cuda.Event recorded by the library where the early part is final (enable_overlap)
        self.side_stream = None
        _LIVE_BUCKETS.add(self)

    def enable_overlap(self):
        """Overlap the all-reduce of the early part with the rest of the backward pass: the library records `early_event` on the
        compute stream once point_layer[0..7]'s gradients are final; `allreduce_sum` then runs that part's collective on a side stream
        behind the event while the compute stream finishes the remaining products, and only the small late part (sigma head,
        point_info, dir_info, colour head: 17 % of the bytes) is reduced after the backward.  CUDA/ROCm tensors only."""
        if self.flat.device.type != "cuda":
            return self
        if self.early_event is None:
            self.early_event = torch.cuda.Event()
            self.early_event.record(torch.cuda.current_stream(self.flat.device))  # torch creates the HIP event at its first record
            self.side_stream = torch.cuda.Stream(self.flat.device)
        return self

    @property
    def early_event_handle(self):
        """hipEvent_t of `early_event` for the C ABI (0 = no overlap)."""
        return int(self.early_event.cuda_event) if self.early_event is not None else 0

    def consume(self):
        """The gradients in the views have been used (optimizer step, all-reduce, or deliberately dropped): the next backward may
        overwrite them.  Called by `allreduce_sum`, and -- through `consume_buckets_of` -- by `train.FusedAdam.step` / `.zero_grad` for
        every bucket whose views ARE the gradients the optimizer just used, so the plain single-process loop
        ``model.grad_bucket = bucket; loss.backward(); opt.step()`` needs no extra call.  A foreign optimizer (``torch.optim.Adam``) knows
        nothing of buckets: call this after its step."""
        self.pending = False

    def _foreign(self):
        return [(v, p) for v, p in zip(self.views, self.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]

    def pack(self):
        for v, p in self._foreign():
            v.copy_(p.grad)

    def unpack(self):
        for v, p in self._foreign():
            p.grad.copy_(v)

    def allreduce_sum(self, group=None):
        """SUM over the ranks, in place.  With `enable_overlap()` and gradients the kernels wrote straight into the views: the early
        part's collective runs on the side stream behind the library's event (i.e. beside the last weight-gradient products), the
        late part's on the current stream; the current stream then waits for both.  Otherwise ONE collective over the whole buffer."""
        foreign = self._foreign()
        if self.early_event is not None and not foreign and 0 < self.early_numel < self.flat.numel():
            cur = torch.cuda.current_stream(self.flat.device)
            self.side_stream.wait_event(self.early_event)
            with torch.cuda.stream(self.side_stream):
                w = dist.all_reduce(self.flat[: self.early_numel], op=dist.ReduceOp.SUM, group=group, async_op=True)
            dist.all_reduce(self.flat[self.early_numel:], op=dist.ReduceOp.SUM, group=group)
            with torch.cuda.stream(cur):
                w.wait()  # the current stream waits for the side stream's collective
            self.consume()
            return
        self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.unpack()
        self.consume()


def consume_buckets_of(params) -> int:
    """Mark every live GradBucket whose views are the `.grad` of `params` as consumed (the optimizer has used, or zero_grad has dropped,
    the gradients a backward wrote into it).  Returns the number of buckets touched."""
    ptrs = {p.grad.data_ptr() for p in params if p.grad is not None}
    n = 0
    for b in list(_LIVE_BUCKETS):
        if any(bp is p for bp, p in zip(b.params, params)) and (not ptrs or any(v.data_ptr() in ptrs for v in b.views)):
            b.consume()
            n += 1
    return n


def train_step_local(model, bucket: GradBucket, row, col, poses_bound, K_inv, C_true, ray0, world: int, group=None):
    """One data-parallel train step on tensors that ARE this rank's slice of the global batch (`NeRFRunner`: the device sampler gathers
    only the slice; `ray0` = (near, far) of the GLOBAL batch's ray 0, host floats -- quirk Q6).  `model.batch_ray` must equal the slice
    size.  Forward + loss + backward with the gradients written straight into `bucket`, then ONE flat SUM all-reduce (in two overlapped
    parts with `bucket.enable_overlap()`), placed where the reference has ``loss.backward(); optimizer.step()`` (nerf.py:473-474).
    Returns this rank's (C_coarse, C_fine, local loss); every rank then holds the full-batch gradient in p.grad (views of `bucket.flat`)."""
    prev_ray0, prev_bucket = model.ray0_near_far, model.grad_bucket
    # `model.check_resample` (mimic nerf.py:251-253: raise where the reference exit(0)s) is a RANK-LOCAL raise; taken before the
    # collective it would leave the other ranks blocked in all_reduce.  The check is therefore made AFTER the all-reduce, its
    # outcome MAX-reduced, and every rank raises together.
    check = bool(getattr(model, "check_resample", False))
    model.ray0_near_far = ray0
    model.grad_bucket = bucket
    try:
        if check:
            model.check_resample = False
        if hasattr(model, "train_step"):  # forward + loss + backward in ONE library call (same kernels, same bits)
            C_c, C_f, loss = model.train_step(row, col, poses_bound, K_inv, C_true)
        else:
            C_c, C_f = model(row, col, poses_bound, K_inv)
            loss = model.ray_loss(C_c, C_f, C_true)
            loss.backward()
    finally:
        model.ray0_near_far, model.grad_bucket = prev_ray0, prev_bucket
        if check:
            model.check_resample = True
    grouped = world > 1 or (dist.is_available() and dist.is_initialized())
    if grouped:
        bucket.allreduce_sum(group)
    else:
        bucket.consume()  # a single process without a group: p.grad (the views) go straight to the optimizer
    if check:
        fault = bool(model.resample_fault())
        if grouped:
            fault = allreduce_host_scalars([1.0 if fault else 0.0], dist.ReduceOp.MAX, row.device if row.device.type == "cuda" else loss.device,
                                           group)[0] > 0.0
        if fault:
            from .nerf import ResampleIndexError

            raise ResampleIndexError("resample index outside [0, Nf-1] on at least one rank (the reference exit(0)s here, nerf.py:251-253)")
    return C_c, C_f, loss


def train_step_sharded(model, bucket: GradBucket, row, col, poses_bound, K_inv, C_true, rank: int, world: int, group=None):
    """`train_step_local` for callers that hold the GLOBAL batch on every rank (every rank passes the same global tensors): this rank's
    contiguous slice is cut here and the global ray 0's (near, far) read from the global `poses_bound`."""
    lo, hi = shard_bounds(row.shape[0], rank, world)
    return train_step_local(model, bucket, row[lo:hi], col[lo:hi], poses_bound[lo:hi], K_inv, C_true[lo:hi], global_ray0(poses_bound),
                            world, group)


class DistEnv:
    """What a launcher (``python -m torch.distributed.run`` / torchrun) tells a rank, read from the ENVIRONMENT only -- no torch.cuda / HIP
    call is involved, so it can (and must) be evaluated before anything touches the GPU (MI355X pool rule: nothing re-execs after that)."""

    def __init__(self, rank: int = 0, world: int = 1, local_rank: int = 0, launched: bool = False):
        self.rank, self.world, self.local_rank, self.launched = rank, world, local_rank, launched

    @classmethod
    def from_env(cls, env=None):
        import os

        env = os.environ if env is None else env
        if "RANK" not in env or "WORLD_SIZE" not in env:
            return cls()
        rank, world = int(env["RANK"]), int(env["WORLD_SIZE"])
        if not 0 <= rank < world:
            raise ValueError(f"RANK={rank} outside WORLD_SIZE={world}")
        return cls(rank, world, int(env.get("LOCAL_RANK", rank)), True)

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def __repr__(self):
        return f"DistEnv(rank={self.rank}, world={self.world}, local_rank={self.local_rank}, launched={self.launched})"


def broadcast_parameters(params, src: int = 0, group=None):
    """Every rank starts from rank `src`'s weights (ONE flat broadcast; replicated weights are what makes the SUM of the slices'
    gradients the full-batch gradient on every rank)."""
    params = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()


def allreduce_host_scalars(values, op, device, group=None):
    """A few host floats through ONE collective (the logging point of the data-parallel loop: loss SUM, fault flag MAX)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op, group=group)
    return t.cpu().tolist()


def batch_shard_bounds(n: int, batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous ray range [lo, hi) of `rank` when n rays are dealt out in whole batches of `batch` rays (the last batch may
    be short): shards start on the batch grid of the reference's display loop (nerf.py:503-520, one DataLoader batch =
    rays [g*batch, (g+1)*batch))."""
    nb = (n + batch - 1) // batch
    lo_b, hi_b = shard_bounds(nb, rank, world)
    return min(lo_b * batch, n), min(hi_b * batch, n)


def render_rows_sharded(model, row, col, poses_bound, K_inv, rank: int, world: int, out: torch.Tensor | None = None,
                        align_to_batches: bool = True, fuse_rays: int = 16384):
    """Inference over a long list of rays (e.g. one frame): this rank renders a contiguous range [lo, hi) in batches of
    model.batch_ray.  No collective.  Returns (lo, hi, C_fine[hi-lo, 3]).

    The reference renders the list in DataLoader batches [g*Bm, (g+1)*Bm) and its resampler takes the coarse spacing from
    ray 0 OF EACH BATCH (nerf.py:233, quirk Q6).  To return the same pixels for any sharding, every kernel call here covers
    rays of reference batches that agree in their ray 0's (near, far) and is handed that pair (`nerf.fuse_plan`; `fuse_rays` = the
    longest call, `model.batch_ray` = one call per batch as up to round 2); with `align_to_batches` (default) shards start
    on the batch grid.  The reference silently drops the tail batch (nerf.py:442); here it is rendered, with its own ray 0."""
    n, Bm = row.shape[0], model.batch_ray
    lo, hi = batch_shard_bounds(n, Bm, rank, world) if align_to_batches else shard_bounds(n, rank, world)
    # NeRFModel.render: the batches of this range that share their ray 0's (near, far) share kernel calls (same bits per ray)
    C = model.render(row, col, poses_bound, K_inv, lo, hi, fuse_rays=fuse_rays)[1] if hi > lo else torch.empty(0, 3)
    if out is not None and hi > lo:
        out[lo:hi] = C
    return lo, hi, C


def gather_rows(C_local: torch.Tensor, n_total: int, rank: int, world: int, group=None, bounds=None, batch: int | None = None) -> torch.Tensor:
    """Assemble the full [n_total, 3] picture on every rank (outside the timed data path).  The shards must be the ones the rows
    were rendered with: `batch` = model.batch_ray for the output of `render_rows_sharded` with its default `align_to_batches`
    (shards on the reference's batch grid, `batch_shard_bounds`); `bounds` = an explicit list of every rank's (lo, hi); neither =
    plain `shard_bounds`.  A local shard of another length than this rank's entry is an error, never silently misplaced."""
    if bounds is not None and batch is not None:
        raise ValueError("gather_rows: give `bounds` or `batch`, not both")
    if bounds is not None:
        sizes = list(bounds)
    elif batch is not None:
        sizes = [batch_shard_bounds(n_total, batch, r, world) for r in range(world)]
    else:
        sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    if C_local.shape[0] != hi - lo:
        raise ValueError(f"gather_rows: rank {rank} holds {C_local.shape[0]} rows but its shard [{lo}, {hi}) has {hi - lo}: pass the "
                         "`batch` (or `bounds`) the rows were rendered with")
    if world == 1:
        return C_local
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros(mx, 3, dtype=C_local.dtype, device=C_local.device)
    pad[: C_local.shape[0]] = C_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)])
