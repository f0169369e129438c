"""One rank of a data-parallel NeRFRunner job, started by tests/test_gpu_parallel.py under ``python -m torch.distributed.run`` in a FRESH
child process (the runner reads RANK / WORLD_SIZE / LOCAL_RANK from the environment before its first GPU call; nothing is re-exec'd).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tests/tools/dp_runner_rank.py OUT_DIR [--bf16] [--split-train] [--iters K] [--batch B] [--force-dist] [--overlap]

N = 1: a real RCCL group (backend "nccl").  N = 2 on the one GPU of a test box: NERF_DIST_BACKEND=gloo (RCCL refuses two ranks on one
device), both ranks on cuda:0 -- the same runner code, the collectives through the host.
Without a launcher (plain ``python dp_runner_rank.py OUT_DIR``): the single-process runner, the thing the ranks are compared with.
Rank 0 writes OUT_DIR/result.pt = {losses, weights (flat), frame, ckpts, ranks}.
"""
import argparse
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--split-train", action="store_true", help="NeRFRunner(split_train=True): the train step in split-fp32 arithmetic")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--force-dist", action="store_true", help="join a process group also as a single rank")
    ap.add_argument("--overlap", action="store_true", help="NeRFRunner(overlap_allreduce=True): early part of the bucket reduced on a side stream")
    args = ap.parse_args()
    if os.environ.get("NERF_DIST_BACKEND") == "gloo":
        os.environ["LOCAL_RANK"] = "0"  # every rank of the rehearsal on the box's one GPU

    import torch

    import nerf_tiny_amd as P

    torch.manual_seed(0)  # the ranks' initial weights are rank 0's anyway (broadcast); the plain run must draw the same ones
    scene = P.data.synthetic_scene(n_pic=3, H=24, W=24, seed=4)
    kw = dict(gpu=0, img_dir="", results_path=os.path.join(args.out, "res") + "/", ckpt_path=os.path.join(args.out, "ck") + "/", low_res=1,
              total_iter=args.iters, batch_ray=args.batch, learning=1e-3, lr_gamma=0.1, lr_milestone=[10, 200], n_coarse=32, n_fine=64,
              data_type="sync", step=args.iters // 2, decay_end=10000, sched="EXP", datasets={"train": scene, "val": scene, "test": scene},
              log_every=1, bf16_mlp=args.bf16, split_train=args.split_train, on_resample_fault="warn")
    run = P.NeRFRunner(continue_=False, distributed=True if args.force_dist else None, overlap_allreduce=True if args.overlap else None, **kw)
    losses = []
    wrote = []
    if run.rank == 0:
        run.writer.add_scalar = lambda tag, v, it: losses.append((tag, float(v), it))
    else:  # only rank 0 has a real writer: the others' is the null writer
        assert type(run.writer).__name__ == "_NullWriter"
    assert run.trainer("train") == args.iters - 1
    frame = run.display(save=True)
    if run.distributed:
        import torch.distributed as dist

        # replicated weights: every rank must hold rank 0's bits after the identical Adam steps on the all-reduced gradients
        flat = torch.cat([p.detach().reshape(-1) for p in run.model.network.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([1.0 if torch.equal(flat, ref) else 0.0], device=flat.device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        assert float(same) == 1.0, "the ranks' weights diverged"
    if run.rank == 0:
        torch.save({"losses": [v for t, v, _ in losses if t.startswith("loss/")],
                    "weights": torch.cat([p.detach().reshape(-1).cpu() for p in run.model.network.parameters()]),
                    "frame": torch.from_numpy(frame), "ckpts": sorted(os.path.basename(f) for f in glob.glob(kw["ckpt_path"] + "*.pkl")),
                    "images": len(glob.glob(kw["results_path"] + "*/*.jpg")), "ranks": run.world, "distributed": run.distributed,
                    "local_rays": run.local_rays}, os.path.join(args.out, "result.pt"))
        print("DP-RUNNER-OK", run.world, flush=True)
    else:
        assert not losses
    if run.distributed:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
