"""Train-loop side of the hot path: the fused optimizer and a runner with the
reference's ``NeRFRunner`` surface (``/root/reference/nerf.py:353-530``) that keeps the device busy -- GPU-resident
ray sampler, fused Adam, no per-iteration host sync (the reference flushes TensorBoard, copies ``C_true`` to the host
and updates a preview image every iteration, nerf.py:478-483).
"""
from __future__ import annotations

import glob
import os
import time

import torch

from . import _abi
from .data import DeviceRays, NeRFDataset
from .nerf import NeRFModel


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (nerf.py:425: betas (0.9, 0.999), eps 1e-7, no weight decay) for the 24 tensors of
    ``model.network`` in ONE kernel launch (``nerf_hip_adam_step``).  A ``torch.optim.Optimizer`` subclass, so the
    reference's ``LambdaLR`` / ``MultiStepLR`` schedulers drive ``param_groups[0]['lr']`` unchanged; ``state_dict`` uses
    Adam's keys (``step``, ``exp_avg``, ``exp_avg_sq``)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(self.param_groups) != 1 or len(ps) != 24:
            raise ValueError("FusedAdam expects one group holding the 24 tensors of NeRFModel.network.parameters()")
        n = sum(p.numel() for p in ps)
        self._m = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        self._v = torch.zeros_like(self._m)
        o = 0
        for p in ps:
            st = self.state[p]
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = self._m[o:o + p.numel()].view_as(p)
            st["exp_avg_sq"] = self._v[o:o + p.numel()].view_as(p)
            o += p.numel()
        self._step = 0

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        ps = self.param_groups[0]["params"]
        o = 0
        for p in ps:  # re-alias the flat buffers
            st = self.state[p]
            self._m[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self._v[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            st["exp_avg"] = self._m[o:o + p.numel()].view_as(p)
            st["exp_avg_sq"] = self._v[o:o + p.numel()].view_as(p)
            o += p.numel()
        self._step = int(self.state[ps[0]]["step"])

    def flat_state(self) -> dict:
        """What a resume needs, flat (parameters() order): {"step", "exp_avg" [593,924], "exp_avg_sq" [593,924]} on the host."""
        return {"step": int(self._step), "exp_avg": self._m.detach().cpu().clone(), "exp_avg_sq": self._v.detach().cpu().clone()}

    def load_flat_state(self, st: dict) -> None:
        if st["exp_avg"].numel() != self._m.numel() or st["exp_avg_sq"].numel() != self._v.numel():
            raise ValueError("optimizer state of another architecture")
        self._m.copy_(st["exp_avg"].to(self._m.device))
        self._v.copy_(st["exp_avg_sq"].to(self._v.device))
        self._step = int(st["step"])
        for p in self.param_groups[0]["params"]:
            self.state[p]["step"] = torch.tensor(float(self._step))

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        ps = g["params"]
        if any(p.grad is None for p in ps):
            raise RuntimeError("FusedAdam.step: every parameter needs a gradient")
        self._step += 1
        b1, b2 = g["betas"]
        dev = ps[0].device
        _abi.check(_abi.lib().nerf_hip_adam_step(_abi.ptr_array(ps), _abi.ptr_array([p.grad.contiguous() for p in ps]),
                                                 self._m.data_ptr(), self._v.data_ptr(), self._step, float(g["lr"]), float(b1),
                                                 float(b2), float(g["eps"]), torch.cuda.current_stream(dev).cuda_stream))
        for p in ps:
            self.state[p]["step"] = torch.tensor(float(self._step))
        # bucket mode (model.grad_bucket: p.grad ARE views of a parallel.GradBucket): the gradients are used up, the next backward may
        # overwrite them
        from .parallel import consume_buckets_of

        consume_buckets_of(ps)
        return None

    def zero_grad(self, set_to_none: bool = True):
        """``Optimizer.zero_grad`` + bucket mode: dropping the gradients also releases the bucket they live in."""
        from .parallel import consume_buckets_of

        consume_buckets_of(self.param_groups[0]["params"])
        super().zero_grad(set_to_none=set_to_none)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def flush(self):
        pass


def _writer():
    try:
        from torch.utils.tensorboard import SummaryWriter  # absent offline

        return SummaryWriter()
    except Exception:
        return _NullWriter()


class NeRFRunner:
    """The reference's runner surface: same 17 constructor arguments (nerf.py:354-372), ``trainer(mode)`` (nerf.py:445)
    and ``display()`` (nerf.py:503).  ``mode`` defaults to "train" so the reference's ``main.py:55`` call works.
    Extra keyword-only arguments: ``datasets`` (dict mode -> dataset, to run without files), ``log_every`` (host sync
    period; the reference syncs every iteration), ``bf16_mlp`` (BASELINE.json cfg3: the MLP on bf16 MFMA, default off), ``split_mlp`` (rendering calls -- validation,
    ``display()`` -- on the split-fp32 inference kernels: same 1e-4 bar, 3x the rate; training forwards ignore it; default off), ``split_train`` (the TRAIN
    step in split-fp32 arithmetic -- forward, dX chain and weight-gradient products on bf16 MFMA with two-part operands: 2x the exact fp32 step, gradients
    inside the bands the exact path is held to; default off, ignored with ``bf16_mlp``),
    ``on_resample_fault`` ("raise" | "warn" | "ignore": what to do when ANY iteration since the last logging point met the reference's
    exit(0) condition of nerf.py:251-253 -- a training run that has died; the kernels record it in a sticky status word, the device path
    itself clamps the index and goes on.  Default "raise": the reference stops there too), ``distributed`` (None: data-parallel iff a
    launcher started more than one rank; True: also with one rank -- rehearsals; False: never), ``overlap_allreduce`` (None: the
    NERF_DP_OVERLAP environment variable, default off; True: reduce the early 83 % of the gradient bucket on a side stream beside the last
    weight-gradient products).

    **Data-parallel training (BASELINE.json cfg3) and tile-sharded rendering (cfg5)**: under ``python -m torch.distributed.run
    --nproc-per-node N .../main.py`` every rank builds this runner on ``cuda:LOCAL_RANK`` (RANK / WORLD_SIZE / LOCAL_RANK are read from
    the environment before any GPU call), joins one RCCL group (backend "nccl"), starts from rank 0's weights and draws the SAME
    permutation from the same sampler seed; of every global ``batch_ray`` batch a rank gathers only its contiguous slice
    (``parallel.shard_bounds``; the global ray 0's near / far go to every rank: quirk Q6), runs forward + loss + backward with the
    gradients written straight into one flat bucket, SUM-all-reduces it (one collective behind the step; optionally the early 83 % beside
    the last weight-gradient products) where the reference has ``loss.backward(); optimizer.step()`` (nerf.py:473-474) and takes the identical fused-Adam step.  Rank 0
    alone logs and writes checkpoints; ``display()`` deals the frames' reference batches out over the ranks with no collective in the
    data path and gathers the pixels afterwards."""

    def __init__(self, gpu=0, img_dir="../nerf_synthetic/lego/", results_path="./results/", ckpt_path="./checkpoint/", low_res=1,
                 total_iter=100000, batch_ray=400, learning=1e-3, lr_gamma=0.1, lr_milestone=(10, 200), n_coarse=64, n_fine=128,
                 data_type="sync", step=100, decay_end=200000, sched="EXP", continue_=False, *, datasets=None, log_every=None,
                 seed=624, bf16_mlp=False, split_mlp=False, split_train=False, on_resample_fault="raise", distributed=None, overlap_allreduce=None):
        from . import nerf as _nerf
        from . import parallel as par

        # ---- the launcher's environment first: nothing above this line has touched the GPU
        self.env = par.DistEnv.from_env()
        self.distributed = (self.env.world > 1) if distributed is None else bool(distributed)
        if not self.distributed:
            self.env = par.DistEnv()  # a stray RANK in the environment of a deliberately single-process run
        self.rank, self.world = self.env.rank, self.env.world
        if not torch.cuda.is_available():
            raise RuntimeError("NeRFRunner needs a ROCm device: the MI355X path has no CPU fallback")
        # one process per GPU: under a launcher the rank's device is LOCAL_RANK; the ini's GPU key is the single-process choice
        self.device = torch.device("cuda:" + str(self.env.local_rank if self.env.launched and self.distributed else gpu))
        _nerf.device = self.device  # module global like nerf.py:387
        torch.cuda.set_device(self.device)
        self.group = None
        if self.distributed:
            import torch.distributed as dist

            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                # "nccl" IS RCCL on ROCm (xGMI).  NERF_DIST_BACKEND=gloo: rehearsals with several ranks on ONE GPU (RCCL refuses two
                # ranks on one device); gloo moves device tensors through the host, same collectives, same results
                backend = os.environ.get("NERF_DIST_BACKEND", "nccl")
                kw_pg = {"device_id": self.device} if backend == "nccl" else {}
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw_pg)
        self.writer = _writer() if self.rank == 0 else _NullWriter()
        self.start_time = time.strftime("%m-%d-%H-%M-%S", time.localtime())
        if self.distributed:  # one name for the job's checkpoints / results: rank 0's clock
            import torch.distributed as dist

            box = [self.start_time]
            dist.broadcast_object_list(box, src=0)
            self.start_time = box[0]
        self.results_path, self.ckpt_path, self.low_res = results_path, ckpt_path, low_res
        self.total_iter, self.batch_ray, self.step, self.decay_end = total_iter, batch_ray, step, decay_end
        self.log_every = log_every or step
        if on_resample_fault not in ("warn", "raise", "ignore"):
            raise ValueError("on_resample_fault: 'warn', 'raise' or 'ignore'")
        self.on_resample_fault = on_resample_fault
        self.resample_fault_iter = None  # first logging point at which the reference's exit(0) condition had been met since the previous one
        # this rank's slice of every global batch: the model is built for the slice (its kernels see B = local rays)
        self.lo, self.hi = par.shard_bounds(batch_ray, self.rank, self.world)
        self.local_rays = self.hi - self.lo
        if self.local_rays < 2:
            raise ValueError(f"BATCH_RAY = {batch_ray} over {self.world} ranks leaves rank {self.rank} {self.local_rays} ray(s): the path needs >= 2 per rank")
        self.model = NeRFModel(num_coarse=n_coarse, num_fine=n_fine, batch_ray=self.local_rays).to(self.device)

        # resume: newest "<anything>_<iter>.pkl" (nerf.py:404-415); every rank reads the same file (one node, one filesystem)
        last_iter, last_ckpt = -1, None
        if continue_:
            for f in glob.glob(ckpt_path + "*.pkl"):
                it = int(f.split("_")[-1][:-4])
                if it > last_iter:
                    last_iter, last_ckpt = it, f
        if last_ckpt is not None:
            self.model = torch.load(last_ckpt, weights_only=False, map_location=self.device).to(self.device)
            self.model.batch_ray = self.local_rays  # (the checkpoint holds the CONFIGURED batch size; this rank's kernels see its slice)
        self.last_iter = last_iter
        self.model.bf16_mlp = bool(bf16_mlp)  # an attribute, not part of the checkpoint format: set after a resume too
        self.model.split_mlp = bool(split_mlp)
        self.model.split_train = bool(split_train)  # opt-in: the train step in split-fp32 arithmetic (2x the exact step's rate)
        # the gradients of every step live in ONE flat buffer whose views are p.grad (the kernels write straight into it: no 24 fresh
        # tensors per iteration) -- the all-reduce buffer of a data-parallel run
        self.bucket = par.GradBucket(self.model.network.parameters())
        if self.distributed:
            par.broadcast_parameters(self.model.network.parameters(), src=0)  # replicated weights, whatever each rank's RNG drew
            # The early 83 % of the bucket (point_layer[0..7]) can be reduced on a side stream beside the last weight-gradient products
            # (GradBucket.enable_overlap).  OPT-IN (overlap_allreduce=True or NERF_DP_OVERLAP=1): measured on a single-rank RCCL group the split
            # launches and the two cross-stream hand-offs cost 46-62 us per step (scripts/dp_step_proxy.py) -- what an 8-rank ring all-reduce
            # of these 2.27 MiB is priced at in the first place -- while ONE collective behind the step costs 1-6 us there.
            if overlap_allreduce if overlap_allreduce is not None else os.environ.get("NERF_DP_OVERLAP") == "1":
                self.bucket.enable_overlap()

        def ds(mode):
            if datasets is not None:
                return datasets[mode]
            return NeRFDataset(root_dir=img_dir, low_res=low_res, transform=None, type=data_type, mode=mode)

        self.train_dataset, self.val_dataset, self.disp_dataset = ds("train"), ds("val"), ds("test")
        # the SAME seeds on every rank: the ranks draw identical permutations and take disjoint slices of every batch
        self.train_rays = DeviceRays(self.train_dataset, self.device, seed)
        self.val_rays = DeviceRays(self.val_dataset, self.device, seed + 1)
        self.disp_rays = DeviceRays(self.disp_dataset, self.device, seed + 2)
        self.height, self.width, self.focal = self.train_dataset.height, self.train_dataset.width, self.train_dataset.focal
        self.num_pic = self.train_dataset.pic_num
        # inverse intrinsics, transposed (nerf.py:433)
        self.K_inv = torch.tensor([[1.0, 0.0, -0.5 * self.width], [0.0, -1.0, 0.5 * self.height], [0.0, 0.0, -self.focal]]).to(torch.float).transpose(0, 1)

        self.optimizer = FusedAdam([{"params": list(self.model.network.parameters()), "initial_lr": learning}], lr=learning,
                                   betas=(0.9, 0.999), eps=1e-7)
        # resume: "<time>_<iter>.opt" beside the checkpoint (written by this runner; absent for a reference-written checkpoint) restores
        # Adam's moments / step and the sampler's position, so that N iterations == k + resume + (N - k) bit for bit in fp32
        self._resume_sampler = None
        if last_ckpt is not None:
            st = self._load_opt_state(last_ckpt)
            if st is not None:
                self._resume_sampler = (st.get("mode", "train"), st["sampler"])
        if sched == "EXP":  # nerf.py:426, including its post-decay_end multiplier lr_gamma * learning
            self.scheduler = torch.optim.lr_scheduler.LambdaLR(
                self.optimizer, lr_lambda=lambda it: lr_gamma ** (it / decay_end) if it < decay_end else lr_gamma * learning,
                last_epoch=self.last_iter)
        else:
            self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, list(lr_milestone), lr_gamma, last_epoch=self.last_iter)

    # ----- the two collectives of the logging point (data-parallel runs only; every rank reaches them at the same iterations) -----
    def _global_loss_and_fault(self, loss, fault: bool, error: bool = False):
        """-> (global loss, any rank's resample fault, any rank's error).  ONE SUM and ONE MAX collective; a rank-local error (a raised
        NerfHipError at the logging point) travels as a flag through the MAX so that EVERY rank raises afterwards -- a rank that left the
        loop alone would leave the others blocked in the next all_reduce until the RCCL timeout."""
        lv = float(loss.detach())
        if not self.distributed:
            return lv, fault, error
        import torch.distributed as dist

        from . import parallel as par

        total = par.allreduce_host_scalars([lv], dist.ReduceOp.SUM, self.device, self.group)[0]  # the loss is a SUM over rays (nerf.py:328-331)
        flags = par.allreduce_host_scalars([1.0 if fault else 0.0, 1.0 if error else 0.0], dist.ReduceOp.MAX, self.device, self.group)
        return total, flags[0] > 0.0, flags[1] > 0.0  # the same decisions on every rank

    # ----- checkpoints: the reference's whole-module pickle (nerf.py:491) + what a bit-exact resume needs beside it -----
    def _save_checkpoint(self, it: int, rays, mode: str, epoch_gen_state, next_batch: int):
        os.makedirs(self.ckpt_path, exist_ok=True)
        stem = self.ckpt_path + self.start_time + "_" + str(it)
        # the module is pickled with the CONFIGURED batch size: a data-parallel rank's model is built for its slice (e.g. 512 of 4096), but
        # the checkpoint must load anywhere and equal a single-process checkpoint of the same config (nerf.py:491, reloaded at :415)
        prev = self.model.batch_ray
        self.model.batch_ray = self.batch_ray
        try:
            torch.save(self.model, stem + ".pkl")
        finally:
            self.model.batch_ray = prev
        # "<time>_<iter>.opt": Adam's moments and step (flat, parameters() order) and the sampler's position -- the reference saves neither
        # (a resumed reference run restarts Adam from zero moments and its DataLoader from a fresh shuffle); absent file = that behaviour
        torch.save({"format": 1, "iter": it, "mode": mode, "adam": self.optimizer.flat_state(),
                    "sampler": {"epoch_gen_state": epoch_gen_state.cpu(), "next_batch": int(next_batch)}}, stem + ".opt")

    def _load_opt_state(self, ckpt: str):
        path = ckpt[:-4] + ".opt"
        if not os.path.exists(path):
            return None
        st = torch.load(path, weights_only=False, map_location="cpu")
        self.optimizer.load_flat_state(st["adam"])
        return st

    # nerf.py:445-499
    def trainer(self, mode="train"):
        from . import parallel as par

        rays = {"train": self.train_rays, "val": self.val_rays, "disp": self.disp_rays}[mode]
        it = self.last_iter + 1
        t0, n0 = time.perf_counter(), it
        self.model.train()
        skip = 0
        if self._resume_sampler is not None and self._resume_sampler[0] == mode:
            # continue the interrupted epoch: the generator state its permutation was drawn from, and the batches already used
            rays.gen.set_state(self._resume_sampler[1]["epoch_gen_state"].cpu())
            skip = int(self._resume_sampler[1]["next_batch"])
        self._resume_sampler = None
        while it < self.total_iter:
            epoch_gen_state = rays.gen.get_state()  # (a host copy of the generator's 16 bytes of Philox state: no device sync)
            batches = (rays.epoch_sharded(self.batch_ray, self.rank, self.world, first_batch=skip) if self.distributed
                       else rays.epoch(self.batch_ray, first_batch=skip))
            bi = skip - 1
            skip = 0
            for batch in batches:
                bi += 1
                row, col, pix_val, poses_bound, pic = batch[:5]
                # (no zero_grad: the backward kernels OVERWRITE the bucket's views, which ARE p.grad; FusedAdam.step releases the bucket)
                if self.distributed:
                    # this rank's slice of the batch: forward + loss + backward into the flat bucket, ONE SUM all-reduce (nerf.py:473-474)
                    _, _, loss = par.train_step_local(self.model, self.bucket, row, col, poses_bound, self.K_inv, pix_val, batch[5],
                                                      self.world, self.group)
                else:
                    # nerf.py:470-473 (forward, ray_loss, backward) as ONE library call: same kernels, no interpreter between them
                    self.model.grad_bucket = self.bucket
                    try:
                        _, _, loss = self.model.train_step(row, col, poses_bound, self.K_inv, pix_val)
                    finally:
                        self.model.grad_bucket = None
                self.optimizer.step()
                self.scheduler.step()
                if (it + 1) % self.log_every == 0:  # the only host sync of the loop
                    # the reference checks its resampling indices in EVERY forward and exit(0)s when a ray's coarse weights have all
                    # vanished (nerf.py:251-253) -- the state a run that has died stays in.  The kernels keep the bit in a sticky word
                    # across iterations; it is looked at here, where the loop syncs anyway, so nothing between two logs is missed
                    err = None
                    try:
                        fault = self.on_resample_fault != "ignore" and self.model.resample_fault_since(clear=True)
                    except _abi.NerfHipError as e:  # rank-local (e.g. STATUS_PREP_TIMEOUT): raised on EVERY rank behind the collectives below
                        err, fault = e, False
                    lv, fault, any_err = self._global_loss_and_fault(loss, fault, err is not None)
                    if any_err:
                        raise err if err is not None else _abi.NerfHipError(f"iteration <= {it}: another rank reported a library error at this logging point")
                    dt = time.perf_counter() - t0
                    self.writer.add_scalar("loss/" + mode, lv, it)
                    self.writer.add_scalar("lr/" + mode, self.optimizer.param_groups[0]["lr"], it)
                    self.writer.flush()
                    if self.rank == 0:
                        print(f"[ITER] {it} [LOSS] {lv:.4f} [LR] {self.optimizer.param_groups[0]['lr']:.3e} "
                              f"[{(it + 1 - n0) * self.batch_ray / max(dt, 1e-9):,.0f} rays/s" + (f", {self.world} ranks]" if self.distributed else "]"))
                    if fault:
                        if self.resample_fault_iter is None:
                            self.resample_fault_iter = it
                            if self.rank == 0:
                                print(f"[ITER] {it} resample index out of range since the last log: the reference prints its banner and exit(0)s "
                                      "there (nerf.py:251-253); this path clamps the index" +
                                      (" and trains on (on_resample_fault='warn')" if self.on_resample_fault == "warn" else ""))
                        if self.on_resample_fault == "raise":
                            from .nerf import ResampleIndexError

                            raise ResampleIndexError(f"iteration <= {it}: resample index outside [0, Nf-1] (the reference exit(0)s here, nerf.py:251-253); "
                                                     "NeRFRunner(on_resample_fault='warn') trains on")
                if (it + 1) % self.step == 0 and self.rank == 0:
                    self._save_checkpoint(it, rays, mode, epoch_gen_state, bi + 1)
                it += 1
                if it >= self.total_iter:
                    break
            if mode == "val":
                break
        self.last_iter = it - 1
        return self.last_iter

    # nerf.py:503-530
    def display(self, save=True):
        rays = self.disp_rays
        result = torch.full((rays.pic_num, self.height, self.width, 3), 1.0, device=self.device)
        self.model.eval()
        # the reference's loop (batches of batch_ray rays in pixel order, the tail < batch stays white) through NeRFModel.render: batches
        # whose ray 0 has the same near / far share kernel calls -- same bits per pixel, launches of up to 16,384 rays instead of 400
        n_keep = rays.num_pix // self.batch_ray * self.batch_ray
        chunk = max(1, (1 << 20) // self.batch_ray) * self.batch_ray  # rays gathered per step (on the batch grid)
        from . import parallel as par

        prev_batch = self.model.batch_ray
        self.model.batch_ray = self.batch_ray  # the grid of the reference's display batches (a data-parallel rank's model is built for its slice)
        try:
            with torch.no_grad():
                for s in range(0, n_keep, chunk):
                    row, col, pix_val, poses_bound, pic = rays.gather(torch.arange(s, min(s + chunk, n_keep), device=self.device))
                    if self.distributed:
                        # cfg5: the chunk's reference batches dealt out over the ranks (shards on the batch grid, every call handed its
                        # batches' ray 0: same pixels for any world size), NO collective in the data path; the picture is assembled after
                        _, _, C_loc = par.render_rows_sharded(self.model, row, col, poses_bound, self.K_inv, self.rank, self.world)
                        C_fine = par.gather_rows(C_loc.to(self.device), row.shape[0], self.rank, self.world, self.group, batch=self.batch_ray)
                    else:
                        _, C_fine = self.model.render(row, col, poses_bound, self.K_inv)
                    result[pic, row, col] = C_fine
        finally:
            self.model.batch_ray = prev_batch
        result = result.cpu().numpy()
        if save and self.rank == 0:
            save_dir = self.results_path + self.start_time + "/"
            os.makedirs(save_dir, exist_ok=True)
            try:
                import matplotlib.pyplot as plt

                for i in range(rays.pic_num):
                    plt.imsave(save_dir + str(i) + ".jpg", result[i].clip(0, 1))
            except Exception as e:  # pragma: no cover
                print("display: could not write images:", e)
            try:
                import imageio
                import numpy as np

                imageio.mimwrite(self.results_path + self.start_time + "_" + str(self.last_iter) + ".mp4",
                                 (result * 255.0).astype(np.uint8), fps=30)
            except Exception:
                pass  # imageio is optional (absent offline)
        return result
