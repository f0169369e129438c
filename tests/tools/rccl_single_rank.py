"""Child process of tests/test_gpu_parallel.py: the data-parallel train step over a REAL RCCL process group (backend "nccl"
on ROCm) with one rank on cuda:0.  Exercises what the multi-GPU job does per rank: GradBucket on device tensors, backward
kernels writing straight into the flat buffer, the SUM all-reduce through RCCL, parallel.train_step_sharded.

    python tests/tools/rccl_single_rank.py          (prints "RCCL-OK ..." and exits 0)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch
import torch.distributed as dist

import nerf_oracle as O
import nerf_tiny_amd as P
from nerf_tiny_amd import parallel as par


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    B, Nc, Nf = 512, 64, 128
    row, col, pb, K, Ct = O.fern_inputs(B, seed=21)
    w = O.make_weights(7, sharp=True)
    row, col, pbd, Ctd = row.to(dev), col.to(dev), pb.to(dev), Ct.to(dev)

    def model(n):
        m = P.NeRFModel(Nc, Nf, n)
        m.load_state_dict(w)
        return m.to(dev)

    def same(a, b):  # summation order of the slab / atomic reductions may differ between two runs: equal to 1e-5, not bitwise
        return float((a.double() - b.double()).norm()) <= 1e-5 * float(b.double().norm()) + 1e-30

    # (a) plain autograd path, no bucket
    m = model(B)
    Cc, Cf = m(row, col, pbd, K)
    m.ray_loss(Cc, Cf, Ctd).backward()
    ref = [p.grad.detach().clone() for p in m.network.parameters()]

    # (b) the sharded step on the full batch (world 1): gradients land in the flat bucket, go through RCCL, stay bit-identical
    m1 = model(B)
    bucket = par.GradBucket(m1.network.parameters())
    par.train_step_sharded(m1, bucket, row, col, pbd, K, Ctd, rank=0, world=1)
    torch.cuda.synchronize()
    for p, v, r in zip(m1.network.parameters(), bucket.views, ref):
        assert p.grad.data_ptr() == v.data_ptr(), "p.grad must alias the flat all-reduce buffer"
        assert same(p.grad, r), "bucketed gradient differs from the plain autograd gradient"
    used = torch.zeros_like(bucket.flat, dtype=torch.bool)
    for v in bucket.views:
        used[(v.data_ptr() - bucket.flat.data_ptr()) // 4:][: v.numel()] = True
    assert int(used.sum()) == 593924 and float(bucket.flat[~used].abs().max()) == 0.0
    assert m1.grad_bucket is None and m1.ray0_near_far is None  # restored

    # a second step overwrites (does not accumulate) -- the C ABI's semantics, like zero_grad() + backward()
    par.train_step_sharded(m1, bucket, row, col, pbd, K, Ctd, rank=0, world=1)
    for p, r in zip(m1.network.parameters(), ref):
        assert same(p.grad, r)

    # (c) two shards of the same batch, each through train_step_sharded (the all-reduce of this 1-rank group is the identity):
    # their flat buffers add up to the full-batch gradient (loss = SUM over rays, global ray 0 handed to both)
    m2 = model(B // 2)
    flats = []
    for r in range(2):
        b2 = par.GradBucket(m2.network.parameters())
        par.train_step_sharded(m2, b2, row, col, pbd, K, Ctd, rank=r, world=2)
        flats.append(b2)
    for x, y, f in zip(flats[0].views, flats[1].views, ref):
        scale = float(x.double().norm() + y.double().norm())
        assert float((x.double() + y.double() - f.double()).norm()) < 2e-5 * scale

    # (d) the bf16-MLP variant takes the same route
    m1.bf16_mlp = True
    par.train_step_sharded(m1, bucket, row, col, pbd, K, Ctd, rank=0, world=1)
    assert all(torch.isfinite(p.grad).all() for p in m1.network.parameters())
    assert not torch.equal(m1.network.point_info.weight.grad, ref[18])

    # (d2) the overlapped collective: point_layer[0..7]'s part of the buffer is reduced on a side stream behind the library's
    # event (nerf_hip_backward_overlap), the rest behind the call -- same gradients as the single collective, fp32 and bf16
    for bf in (False, True):
        m3 = model(B)
        m3.bf16_mlp = bf
        b_plain = par.GradBucket(m3.network.parameters())
        par.train_step_sharded(m3, b_plain, row, col, pbd, K, Ctd, rank=0, world=1)
        torch.cuda.synchronize()
        want = b_plain.flat.clone()
        b_ov = par.GradBucket(m3.network.parameters()).enable_overlap()
        assert b_ov.early_event_handle != 0 and 0 < b_ov.early_numel < b_ov.flat.numel()
        n_early = sum(p.numel() for p in list(m3.network.parameters())[:16])
        assert n_early == 491520 and b_ov.early_numel >= n_early
        for _ in range(3):  # repeated steps: the event and the side stream are reused
            par.train_step_sharded(m3, b_ov, row, col, pbd, K, Ctd, rank=0, world=1)
        torch.cuda.synchronize()
        assert same(b_ov.flat, want), f"overlapped all-reduce changed the gradients (bf16={bf})"
        assert not b_ov.pending

    # (d3) ORDERING of the overlap: with one rank the all-reduce is the identity, so an event recorded too early would
    # still leave the right final buffer.  Check the event itself: poison the buffer, run forward + backward by hand with the bucket's
    # event, and on the side stream BEHIND the event copy the early part away while the compute stream is still busy with the remaining
    # products.  The copy must already be the final early part, bit for bit (a product of point_layer[0..7] moved behind the event, or a
    # reduce that was split differently, would leave poison or partial sums in it).
    for bf in (False, True):
        m4 = model(B)
        m4.bf16_mlp = bf
        b4 = par.GradBucket(m4.network.parameters()).enable_overlap()
        for rep in range(2):
            b4.flat.fill_(float("nan"))
            torch.cuda.synchronize()
            m4.grad_bucket = b4
            Cc, Cf = m4(row, col, pbd, K)
            m4.ray_loss(Cc, Cf, Ctd).backward()   # nerf_hip_backward_overlap records b4.early_event where tensors 0..15 are final
            snap = torch.empty(b4.early_numel, device=dev)
            b4.side_stream.wait_event(b4.early_event)
            with torch.cuda.stream(b4.side_stream):
                snap.copy_(b4.flat[: b4.early_numel], non_blocking=True)
            torch.cuda.synchronize()
            m4.grad_bucket = None
            b4.consume()
            early_real = torch.cat([v.reshape(-1) for v in b4.views[:16]])
            assert torch.isfinite(early_real).all(), f"early part not written (bf16={bf})"
            assert torch.equal(torch.nan_to_num(snap, nan=-7.0), torch.nan_to_num(b4.flat[: b4.early_numel], nan=-7.0)), \
                f"the early event fired before point_layer[0..7]'s gradients were final (bf16={bf}, rep={rep})"
            assert torch.isfinite(torch.cat([v.reshape(-1) for v in b4.views[16:]])).all()

    # (e) the fused optimizer steps from the bucket views
    opt = P.FusedAdam(list(m1.network.parameters()), lr=1e-3)
    before = m1.network.point_info.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m1.network.point_info.weight)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL-OK single-rank nccl group: bucketed backward, all-reduce, overlapped all-reduce + event ordering, shard sum, bf16, fused Adam")


if __name__ == "__main__":
    main()
