"""CPU, world_size 2 over gloo: the host-side logic of the multi-GPU path (sharding, the flat SUM bucket, assembling
shards).  The kernels themselves need a GPU; their shard-sum property is checked on one GPU in
tests/test_gpu_backward.py::test_sum_of_shard_gradients_equals_full_batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nerf_tiny_amd as P
        from nerf_tiny_amd import parallel as par

        torch.manual_seed(0)
        m = P.NeRFModel(64, 128, 8)
        params = list(m.network.parameters())
        # per-rank "gradients": rank r contributes (r + 1) * g
        gen = torch.Generator().manual_seed(1)
        base = [torch.randn(p.shape, generator=gen) for p in params]
        for p, g in zip(params, base):
            p.grad = (rank + 1) * g.clone()
        b = par.GradBucket(params)
        assert b.numel == 593924 and b.numel <= b.flat.numel() < b.numel + 64 * 24 and all((v.data_ptr() - b.flat.data_ptr()) % 256 == 0 for v in b.views)
        # the early / late split of the buffer (overlapped collective on a GPU): point_layer[0..7] = the first 16 tensors; on the CPU
        # enable_overlap() is a no-op and allreduce_sum stays one collective
        ok0 = b.early_numel == (b.views[16].data_ptr() - b.flat.data_ptr()) // 4 and b.early_numel >= sum(p.numel() for p in params[:16]) == 491520
        ok0 &= b.enable_overlap() is b and b.early_event is None and b.early_event_handle == 0
        b.allreduce_sum()
        tot = sum(range(1, world + 1))
        ok = ok0 and all(torch.allclose(p.grad, tot * g, rtol=1e-6, atol=1e-7) for p, g in zip(params, base))
        # shards cover [0, n) exactly, in order
        n = 1003
        bounds = [par.shard_bounds(n, r, world) for r in range(world)]
        ok &= bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        lo, hi = bounds[rank]
        full = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
        got = par.gather_rows(full[lo:hi].clone(), n, rank, world)
        ok &= torch.equal(got, full)
        # the defaults of render_rows_sharded (shards on the reference's batch grid) composed with gather_rows: the batch must be
        # named, and a shard of the wrong length is an error instead of misplaced rows
        batch = 100
        blo, bhi = par.batch_shard_bounds(n, batch, rank, world)
        ok &= (blo, bhi) == ((0, 600) if rank == 0 else (600, n))
        got = par.gather_rows(full[blo:bhi].clone(), n, rank, world, batch=batch)
        ok &= torch.equal(got, full)
        try:
            par.gather_rows(full[blo:bhi].clone(), n, rank, world)  # plain shard_bounds are [0, 502) / [502, 1003)
            ok = False
        except ValueError:
            pass
        dist.barrier()
        # overwrite protection of the bucket: a consumed bucket accepts a new backward, a pending one does not
        b.pending = True
        b.allreduce_sum()
        ok &= b.pending is False
        pb = torch.zeros(4, 17, dtype=torch.float64)
        pb[0, 15], pb[0, 16] = 2.1, 6.3
        ok &= par.global_ray0(pb) == (float(torch.tensor(2.1, dtype=torch.float32)), float(torch.tensor(6.3, dtype=torch.float32)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucket_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


class _FakeModel:
    """Stands in for NeRFModel in train_step_local: no kernels, a rank-local resample fault on demand."""

    def __init__(self, params, fault):
        self.params, self.fault = params, fault
        self.ray0_near_far, self.grad_bucket, self.check_resample = None, None, True
        self.saw_check_during_step = None

    def train_step(self, row, col, pb, K, Ct):
        self.saw_check_during_step = self.check_resample  # a rank-local raise inside the step would strand the other ranks
        b = self.grad_bucket
        b.pending = True
        for v in b.views:
            v.fill_(1.0)
        for p, v in zip(self.params, b.views):
            p.grad = v
        return torch.zeros(row.shape[0], 3), torch.zeros(row.shape[0], 3), torch.tensor(1.0)

    def resample_fault(self):
        return self.fault


def _fault_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nerf_tiny_amd as P
        from nerf_tiny_amd import parallel as par

        real = P.NeRFModel(64, 128, 8)
        params = list(real.network.parameters())
        b = par.GradBucket(params)
        m = _FakeModel(params, fault=(rank == 1))  # ONLY rank 1 meets the reference's exit(0) condition
        row = torch.zeros(4, dtype=torch.int64)
        raised = False
        try:
            par.train_step_local(m, b, row, row, torch.zeros(4, 17), torch.eye(3), torch.zeros(4, 3), (2.0, 6.0), world)
        except P.nerf.ResampleIndexError:
            raised = True
        # both ranks went through the gradient all-reduce (every view = 1 + 1) and BOTH raised; the check was off during the step itself
        summed = all(bool((v == float(world)).all()) for v in b.views)
        dist.barrier()  # nobody is stranded in a collective
        q.put((rank, raised and summed and m.saw_check_during_step is False and m.check_resample is True and not b.pending))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rank_local_resample_fault_raises_on_every_rank_world2():
    """a rank-local raise in front of a collective leaves the other ranks blocked until the RCCL timeout.  With
    model.check_resample on, train_step_local makes the check BEHIND the gradient all-reduce, MAX-reduces its outcome and raises on every rank."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_bounds_properties():
    from nerf_tiny_amd import parallel as par

    for n in (0, 1, 7, 4096, 640000, 762048):
        for world in (1, 2, 3, 8):
            b = [par.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1


def test_fuse_plan_follows_the_batch_grid_and_ray0():
    """nerf.fuse_plan (pure host logic of NeRFModel.render): calls cover the range once, never mix batches whose ray 0 differs in
    (near, far), stay within fuse_rays and sit on the reference's batch grid."""
    import importlib

    nerf = importlib.import_module("nerf-tiny_amd").nerf
    Bm, n = 400, 10_150  # 25 full batches + a tail of 150
    nf = [(2.0, 6.0)] * 26
    plan = nerf.fuse_plan(nf, n, Bm, fuse_rays=4096)
    assert [(s, e) for s, e, _, _ in plan] == [(0, 4000), (4000, 8000), (8000, 10150)]
    assert nerf.fuse_plan(nf, n, Bm, fuse_rays=1) == [(s, min(s + Bm, n), 2.0, 6.0) for s in range(0, n, Bm)]  # never less than a batch
    nf2 = list(nf)
    nf2[7] = (2.0, 6.5)  # one batch whose ray 0 belongs to another picture
    plan = nerf.fuse_plan(nf2, n, Bm, fuse_rays=1 << 20)
    assert [(s, e, fa) for s, e, _, fa in plan] == [(0, 2800, 6.0), (2800, 3200, 6.5), (3200, 10150, 6.0)]
    # a range off the grid: first piece up to the grid, ray 0 of the batch it lies in
    plan = nerf.fuse_plan(nf2, n, Bm, lo=2900, hi=3300, fuse_rays=1 << 20)
    assert plan == [(2900, 3200, 2.0, 6.5), (3200, 3300, 2.0, 6.0)]
    for lo, hi in ((0, n), (123, 9999), (2801, 2802)):
        p = nerf.fuse_plan(nf2, n, Bm, lo=lo, hi=hi, fuse_rays=2000)
        assert p[0][0] == lo and p[-1][1] == hi and all(a[1] == b[0] for a, b in zip(p, p[1:])) and all(e - s <= 2000 for s, e, _, _ in p)
        assert all(nf2[s // Bm] == nf2[(e - 1) // Bm] == (a, b) for s, e, a, b in p)


def test_fuse_plan_can_emit_one_ray_pieces():
    """The plans that hold a 1-ray piece: NeRFModel.render pads such a piece to the library's minimum of two rays
    (nerf.MIN_CALL_RAYS) by repeating its last ray and crops the result; here: the host logic that produces them."""
    import importlib

    nerf = importlib.import_module("nerf-tiny_amd").nerf
    assert nerf.MIN_CALL_RAYS == 2
    # n % batch == 1 and the tail's ray 0 belongs to another picture
    nf = [(2.0, 6.0)] * 4 + [(2.5, 5.5)]
    assert nerf.fuse_plan(nf, 1001, 250)[-1] == (1000, 1001, 2.5, 5.5)
    # a fuse_rays limit that ends one ray before the end
    assert nerf.fuse_plan([(2.0, 6.0)] * 3, 801, 400, fuse_rays=800) == [(0, 800, 2.0, 6.0), (800, 801, 2.0, 6.0)]
    # an unaligned shard that starts on the last ray of a batch whose successor has another near / far
    nf = [(2.0, 6.0), (2.0, 6.0), (2.5, 5.5), (2.5, 5.5)]
    plan = nerf.fuse_plan(nf, 1600, 400, lo=799, hi=1200)
    assert plan == [(799, 800, 2.0, 6.0), (800, 1200, 2.5, 5.5)]


def test_consume_buckets_of_releases_the_bucket_the_optimizer_used():
    """parallel.consume_buckets_of (called by train.FusedAdam.step / .zero_grad): a bucket whose views ARE the params' gradients is
    released, a bucket of other parameters or one whose views are not the current gradients is left alone (host logic, CPU tensors)."""
    from nerf_tiny_amd import parallel as par

    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    other = [torch.nn.Parameter(torch.zeros(4))]
    b, b_other = par.GradBucket(ps), par.GradBucket(other)
    for p, v in zip(ps, b.views):
        p.grad = v
    other[0].grad = b_other.views[0]
    b.pending = b_other.pending = True
    assert par.consume_buckets_of(ps) == 1
    assert b.pending is False and b_other.pending is True
    # gradients that live elsewhere (plain autograd): this bucket was not what the optimizer used
    b.pending = True
    for p in ps:
        p.grad = torch.zeros_like(p)
    assert par.consume_buckets_of(ps) == 0 and b.pending is True
    # zero_grad(set_to_none=True) before: no gradient at all -> every bucket of these parameters is released (its content is dropped)
    for p in ps:
        p.grad = None
    assert par.consume_buckets_of(ps) == 1 and b.pending is False


def test_dist_env_is_read_from_the_environment_only():
    from nerf_tiny_amd import parallel as par

    e = par.DistEnv.from_env({})
    assert (e.rank, e.world, e.local_rank, e.launched, e.is_main) == (0, 1, 0, False, True)
    e = par.DistEnv.from_env({"RANK": "5", "WORLD_SIZE": "8", "LOCAL_RANK": "5"})
    assert (e.rank, e.world, e.local_rank, e.launched, e.is_main) == (5, 8, 5, True, False)
    assert par.DistEnv.from_env({"RANK": "1", "WORLD_SIZE": "2"}).local_rank == 1  # a launcher without LOCAL_RANK: one node
    with pytest.raises(ValueError):
        par.DistEnv.from_env({"RANK": "2", "WORLD_SIZE": "2"})


def _cpu_rays(seed):
    """DeviceRays on the CPU with a pure-torch gather (the kernel nerf_hip_gather_rays needs a GPU; its parity is tests/test_gpu_train.py's):
    the sampler's HOST logic -- permutation, slices, drop_last, the global ray 0 -- is what the data-parallel runner builds on."""
    import nerf_tiny_amd as P

    ds = P.data.synthetic_scene(n_pic=3, H=10, W=14, seed=2)
    ds.poses_bounds[:, 15] = [2.0, 2.5, 3.0]  # per-picture near / far (LLFF-like), so ray 0 matters
    ds.poses_bounds[:, 16] = [6.0, 5.5, 7.0]
    rays = P.data.DeviceRays(ds, "cpu", seed)

    def gather(index):
        pic = index // (rays.height * rays.width)
        rem = index % (rays.height * rays.width)
        return rem // rays.width, rem % rays.width, rays.pixels[index], rays.poses[pic], pic

    rays.gather = gather
    return rays


def _dp_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_tiny_amd import parallel as par

        ok = True
        # (1) same seed on every rank -> same permutation; the slices of a batch are disjoint, contiguous and cover it; ray 0 is the GLOBAL one
        Bm = 50
        mine = list(_cpu_rays(624).epoch_sharded(Bm, rank, world))
        whole = list(_cpu_rays(624).epoch(Bm))
        ok &= len(mine) == len(whole) == (3 * 10 * 14) // Bm
        lo, hi = par.shard_bounds(Bm, rank, world)
        for (row, col, pix, pb, pic, ray0), (frow, fcol, fpix, fpb, fpic) in zip(mine, whole):
            ok &= torch.equal(row, frow[lo:hi]) and torch.equal(col, fcol[lo:hi]) and torch.equal(pix, fpix[lo:hi]) and torch.equal(pb, fpb[lo:hi])
            ok &= ray0 == (float(fpb[0, 15]), float(fpb[0, 16])) == par.global_ray0(fpb)
        # every rank's slices, gathered, rebuild the global batches
        rows = [torch.empty_like(mine[0][0]) for _ in range(world)]
        dist.all_gather(rows, mine[0][0])
        ok &= torch.equal(torch.cat(rows), whole[0][0])
        # (2) replicated weights from rank 0, whatever each rank drew
        torch.manual_seed(100 + rank)
        ps = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
        par.broadcast_parameters(ps, src=0)
        torch.manual_seed(100)
        ok &= torch.equal(ps[0].detach(), torch.randn(7, 3)) and torch.equal(ps[1].detach(), torch.randn(5))
        # (3) the logging point: loss = SUM over ranks, the fault flag = MAX (every rank takes the same decision)
        ok &= par.allreduce_host_scalars([1.5 + rank], dist.ReduceOp.SUM, "cpu") == [sum(1.5 + r for r in range(world))]
        ok &= par.allreduce_host_scalars([1.0 if rank == 1 else 0.0], dist.ReduceOp.MAX, "cpu") == [1.0]
        # (4) the step itself on stand-in gradients: each rank's slice gradient lands in its bucket, the flat SUM is the full-batch gradient
        b = par.GradBucket(ps)
        for p, v in zip(ps, b.views):
            v.fill_(float(rank + 1))
            p.grad = v
        b.pending = True
        b.allreduce_sum()
        ok &= (not b.pending) and all(bool((p.grad == sum(range(1, world + 1))).all()) for p in ps)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_runner_host_logic_world2():
    """The host logic NeRFRunner's data-parallel mode is made of (the runner itself needs a GPU: tests/test_gpu_parallel.py): sampler
    slices and the global ray 0, replicated start weights, the two collectives of the logging point, the bucket's SUM."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
