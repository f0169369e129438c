"""Host-side mirror of the reference's ``nerf`` module surface for the volume-rendering hot path.

Same names, constructor arguments, call signatures and ``state_dict`` keys as the reference
(``/root/reference/nerf.py``): ``NeRFModel(num_coarse, num_fine, batch_ray)`` (nerf.py:170),
``model(row, column, poses_bound, K_inv) -> (C_coarse, C_fine)`` (nerf.py:333-348),
``model.ray_loss`` (nerf.py:325-331), ``model.network.parameters()`` for Adam (nerf.py:425).
The arithmetic happens in libnerf_hip.so (hand-written gfx950 kernels) through a
``torch.autograd.Function``; PyTorch only owns device memory, the stream and autograd plumbing.
There is NO fallback: calling a model that does not live on a ROCm device raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch
import torch.nn as nn

from . import _abi

# module-global device like the reference (nerf.py:40, :387); the model follows its parameters' device.
device = None

LAST_DELTA = 0.0001  # nerf.py:286
MIN_CALL_RAYS = 2     # include/nerf_hip.h: 2 <= B (the reference crashes on B = 1, nerf.py:208)


class Activation(nn.Module):
    """nerf.py:69-74 (abs); kept so pickled/introspected module trees look the same."""

    def forward(self, x):
        return torch.abs(x)


class Network(nn.Module):
    """Parameter container with the reference's module tree (nerf.py:76-99) so that ``state_dict``
    keys, shapes and ``parameters()`` order are identical.  The forward pass is the fused HIP kernel
    (csrc/field_fwd.hip); this module cannot be called on its own."""

    def __init__(self, point_dim=60, dir_dim=24, depth=8, width=256, batch_size=8, layers_skip=[4]):
        super().__init__()
        if (point_dim, dir_dim, depth, width, list(layers_skip)) != (60, 24, 8, 256, [4]):
            raise ValueError("libnerf_hip.so is built for the reference's architecture: 60/24 inputs, 8x256, skip at 4")
        self.depth, self.width, self.batch_size, self.layers_skip = depth, width, batch_size, layers_skip
        self.point_layer = nn.ModuleList([nn.Sequential(nn.Linear(point_dim, width), nn.ReLU(True))])
        for i in range(1, depth):
            fan_in = width + point_dim if i in layers_skip else width
            self.point_layer.append(nn.Sequential(nn.Linear(fan_in, width), nn.ReLU(True)))
        self.sigma_layer = nn.Sequential(nn.Linear(width, 1), Activation())
        self.point_info = nn.Linear(width, width)
        self.dir_info = nn.Sequential(nn.Linear(width + dir_dim, width // 2), nn.ReLU(True))
        self.color_layer = nn.Sequential(nn.Linear(width // 2, 3), nn.Sigmoid())

    def forward(self, *args, **kwargs):
        raise RuntimeError("Network is evaluated inside libnerf_hip.so (NeRFModel.forward); it has no eager path")


class Encoder(nn.Module):
    """nerf.py:126-167: no parameters; the encoding is fused into the field kernel."""

    def __init__(self, L_point=10, L_dir=4, batch_size=8):
        super().__init__()
        if (L_point, L_dir) != (10, 4):
            raise ValueError("libnerf_hip.so is built for L_point=10, L_dir=4")
        self.L_point, self.L_dir, self.batch_size = L_point, L_dir, batch_size

    def forward(self, *args, **kwargs):
        raise RuntimeError("Encoder is fused into libnerf_hip.so (NeRFModel.forward); it has no eager path")


class ResampleIndexError(RuntimeError):
    """Raised (when ``model.check_resample`` is on) where the reference prints a banner and exit(0)s (nerf.py:251-253)."""


def _call_flags(model, need_grad: bool) -> int:
    bf16 = getattr(model, "bf16_mlp", False)
    # split-fp32: `split_mlp` selects it for inference calls; `split_train` (its own opt-in switch) for TRAINING calls -- forward, dX chain and
    # weight-gradient products on bf16 MFMA with two-part operands (csrc/field_fwd_split.hip, field_bwd_split.hip, dw_bf16.hip's SPLIT form)
    split = (getattr(model, "split_train", False) and not model.force_tile_kernel) if need_grad else getattr(model, "split_mlp", False)
    return ((_abi.SAVE_FOR_BACKWARD if need_grad else 0) | (_abi.FORCE_TILE_KERNEL if model.force_tile_kernel else 0)
            | (_abi.BF16_MLP if bf16 else 0) | (_abi.CORRECTED if getattr(model, "corrected", False) else 0)
            | (_abi.SPLIT_MLP if split and not bf16 else 0))


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, need_grad, row, col, pb, K9, ray0, *params):
        B = row.shape[0]
        Nc, Nf = model.num_coarse, model.num_fine
        flags = _call_flags(model, need_grad)
        ws = model._workspace(B, flags)  # (inside NeRFModel.render: the frame's one workspace, sized for its longest call)
        # rendering loops (`with model.frozen_weights():`): the packed weight image a previous call of the SAME frozen section
        # left in this workspace is reused.  Outside such a section the image is rebuilt on every call (12 us): a version
        # stamp cannot see writes through `.data`, dist.broadcast or raw pointers.
        reuse = model._frozen and not need_grad and (ws.data_ptr(), flags) in model._packed
        call_flags = flags | (_abi.WEIGHTS_UNCHANGED if reuse else 0)
        dev = row.device
        C_c = torch.empty(B, 3, dtype=torch.float32, device=dev)
        C_f = torch.empty(B, 3, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        wptr = _abi.ptr_array(params)
        _abi.check(_abi.lib().nerf_hip_forward(wptr, row.data_ptr(), col.data_ptr(), pb.data_ptr(), K9,
                                               ray0, B, Nc, Nf, LAST_DELTA, C_c.data_ptr(), C_f.data_ptr(),
                                               ws.data_ptr(), ws.numel(), call_flags, stream))
        if model._frozen and not need_grad:
            model._packed.add((ws.data_ptr(), flags))
        model._last_ws = ws
        if need_grad:
            gen = model._ws_generation.get(flags, 0) + 1
            model._ws_generation[flags] = gen
            ctx.generation = gen
            ctx.model, ctx.ws, ctx.flags, ctx.B = model, ws, flags, B
            ctx.params, ctx.ray0 = params, ray0
            ctx.bucket = model.grad_bucket  # bound when the graph is recorded: toggling model.grad_bucket later does not change this step
        return C_c, C_f

    @staticmethod
    def backward(ctx, dC_c, dC_f):
        model = ctx.model
        if ctx.generation != model._ws_generation.get(ctx.flags):
            raise RuntimeError("the workspace of this forward was reused by a later training forward; call backward first")
        params = ctx.params
        bucket = ctx.bucket
        if bucket is not None:
            # data-parallel trainer: the kernels write straight into views of the flat all-reduce buffer (parallel.GradBucket).
            # Overwrite semantics: ONE backward per step.  torch.autograd.grad, parameter hooks, gradient accumulation and a second
            # loss through the same model are not supported in bucket mode (they would see None / overwritten gradients).
            if len(bucket.params) != len(params) or any(a is not b for a, b in zip(bucket.params, params)):
                raise RuntimeError("model.grad_bucket was built for other parameters")
            if bucket.pending:
                raise RuntimeError("a second backward would overwrite the gradients of the previous one in model.grad_bucket before "
                                   "they were used: bucket.allreduce_sum(), train.FusedAdam.step() / .zero_grad() release it; after any other "
                                   "optimizer's step call bucket.consume()")
            bucket.pending = True
            grads = bucket.views
        else:
            grads = [torch.empty_like(p) for p in params]
        dC_c = dC_c.contiguous().float()
        dC_f = dC_f.contiguous().float()
        stream = torch.cuda.current_stream(dC_c.device).cuda_stream
        # a bucket with overlap enabled gets the event at which point_layer[0..7]'s gradients are final (parallel.GradBucket)
        early = bucket.early_event_handle if bucket is not None else 0
        _abi.check(_abi.lib().nerf_hip_backward_overlap(_abi.ptr_array(params), dC_c.data_ptr(), dC_f.data_ptr(), ctx.ray0, ctx.B,
                                                        model.num_coarse, model.num_fine, LAST_DELTA, _abi.ptr_array(grads),
                                                        ctx.ws.data_ptr(), ctx.ws.numel(), ctx.flags, stream, early or None))
        if bucket is not None:
            # p.grad IS the bucket view (overwritten every step, like the C ABI's dweights24): nothing for autograd to accumulate
            for p, v in zip(params, grads):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
            return (None,) * (7 + len(params))
        return (None, None, None, None, None, None, None, *grads)


class NeRFModel(nn.Module):
    """Drop-in for the reference's ``NeRFModel`` (nerf.py:169-348)."""

    def __init__(self, num_coarse=64, num_fine=128, batch_ray=8):
        super().__init__()
        self.encoder = Encoder(batch_size=batch_ray)
        self.network = Network(batch_size=batch_ray)
        self.num_coarse = num_coarse
        self.num_fine = num_fine
        self.batch_ray = batch_ray
        #: mimic nerf.py:251-253 (costs one host sync per call, like the reference): raise instead of exit(0)
        self.check_resample = False
        #: (near, far) of the GLOBAL ray 0 when one batch is sharded over several GPUs (quirk Q6); None = local ray 0
        self.ray0_near_far = None
        #: use the LDS-tile field kernels instead of the register-resident ones (A/B measurements, tests)
        self.force_tile_kernel = False
        #: BASELINE.json cfg3: run the MLP on bf16 MFMA (fp32 accumulation, fp32 everything else); ~1e-2 of the fp32 result
        self.bf16_mlp = False
        #: INFERENCE calls (no grad) only: evaluate the fp32 MLP on bf16 MFMA with every fp32 operand split into two bf16 parts (hi + mid)
        #: and three MFMAs per product, fp32 accumulation -- inside the same 1e-4 bar as the exact-fp32 default,
        #: 3x faster.  Off by default: the default keeps exact k-ordered fp32 fma chains; training forwards ignore it (their switch is split_train)
        self.split_mlp = False
        #: TRAINING calls (a forward that records a graph, train_step): the whole train step in split-fp32 arithmetic -- every fp32 operand of
        #: the forward, the dX chain and the weight-gradient products as two bf16 parts (hi + mid), three bf16 MFMAs per product, fp32
        #: accumulation.  Opt-in, never the headline: the default keeps the exact fp32 kernels.  Loss to 1e-5 of the exact path's, gradients
        #: inside the bands the exact path is held to against the reference (tests/test_gpu_split.py)
        self.split_train = False
        #: OPTIONAL EXTRA, off by default, NOT the reference's results (SURVEY.md 8a "Q"): one stable sort of the merged samples by depth with
        #: rgb / sigma moving along (instead of nerf.py:307-308's five independent channel sorts) and a detached t_fine (instead of
        #: nerf.py:259's attached one).  Parity unpinned -- the reference has no such mode; tested against the oracle's restatement only
        self.corrected = False
        #: parallel.GradBucket or None.  When set (read when the forward records the graph), backward writes the 24 gradients straight
        #: into the bucket's flat buffer and makes p.grad its views (overwrite semantics: one backward per step, a second one before
        #: the gradients were consumed raises; autograd.grad / hooks unsupported), so the all-reduce needs no pack / unpack
        self.grad_bucket = None
        self._ws_capacity = False  # True inside render(): an inference slot sized for more rays serves shorter calls too
        self._ws = {}            # flags -> (key, workspace): one slot per flag set (training / inference / bf16 ...)
        self._ws_generation = {}  # flags -> count of training forwards on that slot
        self._last_ws = None
        self._frozen = False
        self._packed = set()

    # ----- plumbing -------------------------------------------------------------------------
    def _workspace(self, B, flags):
        """One workspace per flag set, so that a validate-while-training loop (17 GB training workspace at cfg2 + a small
        inference one) does not reallocate on every switch; a slot is replaced when its sizes or device change."""
        dev = self.network.point_info.weight.device
        key = (B, self.num_coarse, self.num_fine, flags, dev)
        slot = self._ws.get(flags)
        if slot is not None and self._ws_capacity and slot[0][1:] == key[1:] and slot[0][0] >= B and not (flags & _abi.SAVE_FOR_BACKWARD):
            return slot[1]  # NeRFModel.render: calls of different lengths share the slot sized for the longest one (the regions in
            #                 front of the per-ray ones -- status, packed weight images -- do not move with B: csrc/api.hip layout())
        if slot is None or slot[0] != key:
            self._ws.pop(flags, None)
            slot = None
            n = _abi.ws_bytes(B, self.num_coarse, self.num_fine, flags)
            ws = torch.empty(n, dtype=torch.uint8, device=dev)
            assert ws.data_ptr() % 256 == 0
            ws[:256].zero_()  # the status region: its sticky word is the caller's to initialise (nerf_hip_read_status_sticky)
            self._ws[flags] = slot = (key, ws)
            self._packed = {k for k in self._packed if k[1] != flags}
        return slot[1]

    @property
    def last_workspace(self):
        """The workspace the most recent forward ran on (introspection: _abi.ws_view, status word)."""
        return self._last_ws

    @contextlib.contextmanager
    def frozen_weights(self):
        """Rendering loops: inside this section the caller guarantees that no parameter is written, so the packed weight
        image is built once per workspace and reused (NERF_HIP_WEIGHTS_UNCHANGED) instead of rebuilt on every call.
        Opt-in because no stamp can see every write (``p.data.add_()``, ``dist.broadcast(p.data)``, raw pointers)."""
        prev = self._frozen
        self._frozen = True
        if not prev:
            self._packed = set()
        try:
            yield self
        finally:
            self._frozen = prev
            if not prev:
                self._packed = set()

    def __getstate__(self):  # torch.save(model) (nerf.py:491) must not pickle the workspace
        d = dict(self.__dict__)
        d["_ws"] = {}
        d["_ws_generation"] = {}
        d["_last_ws"] = None
        d["_packed"] = set()
        d["_frozen"] = False
        d["_ws_capacity"] = False
        d["grad_bucket"] = None
        return d

    def __setstate__(self, d):  # checkpoints written by an earlier build lack the newer plumbing attributes
        super().__setstate__(d)
        self.__dict__.setdefault("grad_bucket", None)
        self.__dict__.setdefault("split_mlp", False)
        self.__dict__.setdefault("split_train", False)
        self.__dict__.setdefault("corrected", False)
        self.__dict__["_ws"], self.__dict__["_ws_generation"] = {}, {}
        self.__dict__["_last_ws"], self.__dict__["_packed"], self.__dict__["_frozen"] = None, set(), False
        self.__dict__["_ws_capacity"] = False

    def _params(self):
        ps = list(self.network.parameters())
        for p in ps:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise RuntimeError("network parameters must be contiguous fp32")
        return ps

    # ----- reference surface ----------------------------------------------------------------
    def ray_loss(self, C_coarse, C_fine, C_true):
        """nerf.py:325-331: sum (not mean) of squared errors of both colours."""
        return _RayLossFn.apply(C_coarse, C_fine, C_true)

    def forward(self, row, column, poses_bound, K_inv):
        """nerf.py:333-348.  row/column [B] i64, poses_bound [B,17] (any float dtype), K_inv [3,3];
        CPU or device tensors.  Returns (C_coarse, C_fine) [B,3] fp32 on the model's device."""
        ps = self._params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("NeRFModel runs only on a ROCm device (MI355X): model.to('cuda'); there is no CPU path")
        if row.shape[0] != self.batch_ray:
            raise ValueError(f"batch of {row.shape[0]} rays, model built for batch_ray={self.batch_ray} (nerf.py:172-176)")
        return self._launch(ps, row, column, poses_bound, K_inv)

    def _launch(self, ps, row, column, poses_bound, K_inv):
        dev = ps[0].device
        K9 = _abi.f32_array(K_inv.detach().to("cpu", torch.float32).reshape(-1).tolist())
        pb = poses_bound.to(torch.float).to(dev).contiguous()  # cast first like nerf.py:338
        row_d = row.to(dev, torch.int64).contiguous()
        col_d = column.to(dev, torch.int64).contiguous()
        ray0 = _abi.f32_array(self.ray0_near_far) if self.ray0_near_far is not None else None
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in ps)  # grad mode is off inside Function.forward
        C_c, C_f = _RenderFn.apply(self, need_grad, row_d, col_d, pb, K9, ray0, *ps)
        if self.check_resample and self.resample_fault():
            raise ResampleIndexError("resample index outside [0, Nf-1] (the reference exit(0)s here, nerf.py:251-253)")
        return C_c, C_f

    def train_step(self, row, column, poses_bound, K_inv, C_true):
        """The device work of one iteration of the reference's loop (nerf.py:470-473: ``model(...)``, ``ray_loss``, ``loss.backward()``) in ONE
        library call (``nerf_hip_train_step``): the same kernels enqueued back to back without the interpreter between them -- at a 400 / 512-ray
        batch three separate calls leave 3 % of the step in gaps.  Leaves the gradients of this batch in ``p.grad`` exactly as the autograd
        path does -- fresh tensors, or the views of ``model.grad_bucket`` (overwrite semantics, ``pending`` set) -- and returns
        ``(C_coarse, C_fine, loss)`` (detached; ``loss`` a 0-dim tensor).  Bit-identical to the three calls
        (tests/test_gpu_train.py::test_fused_train_step_equals_autograd_path).  What ``NeRFRunner.trainer`` calls."""
        ps = self._params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("NeRFModel runs only on a ROCm device (MI355X): model.to('cuda'); there is no CPU path")
        if row.shape[0] != self.batch_ray:
            raise ValueError(f"batch of {row.shape[0]} rays, model built for batch_ray={self.batch_ray} (nerf.py:172-176)")
        B, Nc, Nf = row.shape[0], self.num_coarse, self.num_fine
        K9 = _abi.f32_array(K_inv.detach().to("cpu", torch.float32).reshape(-1).tolist())
        pb = poses_bound.to(torch.float).to(dev).contiguous()
        row_d, col_d = row.to(dev, torch.int64).contiguous(), column.to(dev, torch.int64).contiguous()
        Ct = C_true.to(dev, torch.float32).contiguous()
        ray0 = _abi.f32_array(self.ray0_near_far) if self.ray0_near_far is not None else None
        flags = _call_flags(self, True)
        ws = self._workspace(B, flags)
        bucket = self.grad_bucket
        if bucket is not None:
            if len(bucket.params) != len(ps) or any(a is not b for a, b in zip(bucket.params, ps)):
                raise RuntimeError("model.grad_bucket was built for other parameters")
            if bucket.pending:
                raise RuntimeError("a second backward would overwrite the gradients of the previous one in model.grad_bucket before "
                                   "they were used: bucket.allreduce_sum(), train.FusedAdam.step() / .zero_grad() release it; after any other "
                                   "optimizer's step call bucket.consume()")
            bucket.pending = True
            grads = bucket.views
        else:
            grads = [torch.empty_like(p) for p in ps]
        C_c = torch.empty(B, 3, dtype=torch.float32, device=dev)
        C_f = torch.empty(B, 3, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        early = bucket.early_event_handle if bucket is not None else 0
        _abi.check(_abi.lib().nerf_hip_train_step(_abi.ptr_array(ps), row_d.data_ptr(), col_d.data_ptr(), pb.data_ptr(), K9, ray0, Ct.data_ptr(),
                                                  B, Nc, Nf, LAST_DELTA, C_c.data_ptr(), C_f.data_ptr(), loss.data_ptr(), _abi.ptr_array(grads),
                                                  ws.data_ptr(), ws.numel(), flags, torch.cuda.current_stream(dev).cuda_stream, early or None))
        self._last_ws = ws
        self._ws_generation[flags] = self._ws_generation.get(flags, 0) + 1  # a graph recorded earlier on this slot is stale now
        for p, g in zip(ps, grads):
            if bucket is not None:
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                    p.grad = g
            else:
                p.grad = g  # overwrite, like zero_grad(set_to_none=True); backward()
        if self.check_resample and self.resample_fault():
            raise ResampleIndexError("resample index outside [0, Nf-1] (the reference exit(0)s here, nerf.py:251-253)")
        return C_c, C_f, loss.reshape(())

    def resample_fault(self) -> bool:
        """Did the most recent forward meet the reference's exit condition -- a ray whose resampling index falls outside [0, Nf-1], i.e.
        whose coarse weights all vanished (nerf.py:251-253: banner + exit(0))?  The device path clamps the index and goes on; this reads
        the status word the kernels left (ONE host sync)."""
        return bool(self.read_status() & _abi.STATUS_RESAMPLE_INDEX)

    def read_status(self) -> int:
        """The status word of the most recent call on this model (``nerf_hip_read_status``: ONE host sync).  Raises ``NerfHipError`` on
        STATUS_PREP_TIMEOUT -- the one-launch preparation of a bf16-MLP call gave up waiting for its weight fold, the packed weight image
        is poisoned with NaN and so are the outputs of every call that used it (csrc/prep_bf16.hip) -- so that no caller renders or trains
        on from there."""
        ws = self._last_ws
        if ws is None:
            return 0
        st = C.c_uint32(0)
        _abi.check(_abi.lib().nerf_hip_read_status(ws.data_ptr(), ws.numel(), C.byref(st), torch.cuda.current_stream(ws.device).cuda_stream))
        if st.value & _abi.STATUS_PREP_TIMEOUT:
            raise _abi.NerfHipError("a bf16-MLP call's one-launch preparation gave up waiting for the weight fold (prep_bf16.hip): the packed weight "
                                    "image is poisoned, the outputs of that call are NaN")
        return int(st.value)

    def resample_fault_since(self, clear: bool = True) -> bool:
        """Did ANY forward on the current workspaces meet that condition since the last call that cleared the record?  The kernels OR the
        status bits into a sticky word no forward resets (``nerf_hip_read_status_sticky``), so a train loop that looks only at its logging
        points misses nothing in between -- the reference checks every forward (nerf.py:251-253).  One host sync per workspace."""
        hit = timeout = False
        for _, ws in self._ws.values():  # every workspace is read (and cleared) before anything is raised
            st = C.c_uint32(0)
            _abi.check(_abi.lib().nerf_hip_read_status_sticky(ws.data_ptr(), ws.numel(), C.byref(st), 1 if clear else 0,
                                                              torch.cuda.current_stream(ws.device).cuda_stream))
            timeout = timeout or bool(st.value & _abi.STATUS_PREP_TIMEOUT)
            hit = hit or bool(st.value & _abi.STATUS_RESAMPLE_INDEX)
        if timeout:
            raise _abi.NerfHipError("a bf16-MLP call's one-launch preparation gave up waiting for the weight fold (prep_bf16.hip): the packed "
                                    "weight image of that call was poisoned, its results are NaN")
        return hit

    @torch.no_grad()
    def render(self, row, column, poses_bound, K_inv, lo: int = 0, hi: int | None = None, fuse_rays: int = 16384):
        """Inference over a LONG list of rays (a frame, a test set) -- rays [lo, hi) of it -- with the reference's batch semantics and few
        kernel calls: the list is the sequence of batches [g*batch_ray, (g+1)*batch_ray) the reference's display loop feeds to `forward`
        (nerf.py:503-520), every ray gets exactly the bits a per-batch `forward` gives it (`fuse_plan`: consecutive batches whose ray 0
        has the same near / far share a launch of up to `fuse_rays` rays), and a 400-ray batch size no longer means 400-ray launches
        (0.71 of the fp32 roof, a fifth of the chip for the bf16 kernels).  The tail batch is rendered too (the reference drops it,
        nerf.py:442), with its own ray 0.  The weights must not change during the call.  Returns (C_coarse, C_fine) [hi - lo, 3]."""
        ps = self._params()
        dev = ps[0].device
        n, Bm = row.shape[0], self.batch_ray
        hi = n if hi is None else hi
        C_c = torch.empty(max(hi - lo, 0), 3, dtype=torch.float32, device=dev)
        C_f = torch.empty_like(C_c)
        if hi <= lo:
            return C_c, C_f
        starts = torch.arange(0, n, Bm)
        nf0 = poses_bound[starts.to(poses_bound.device)][:, 15:17].to(torch.float32).cpu().tolist()  # ONE host copy per call
        plan = fuse_plan(nf0, n, Bm, lo, hi, fuse_rays)
        prev_ray0, prev_cap = self.ray0_near_far, self._ws_capacity
        try:
            self._ws_capacity = True
            self._workspace(max(MIN_CALL_RAYS, max(e - s for s, e, _, _ in plan)), _call_flags(self, False))  # ONE workspace, sized for the longest call
            with self.frozen_weights():
                for s, e, near, far in plan:
                    self.ray0_near_far = (near, far)
                    r_, c_, p_ = row[s:e], column[s:e], poses_bound[s:e]
                    if e - s < MIN_CALL_RAYS:
                        # a 1-ray piece (n % batch_ray == 1 behind a batch with another near / far; an unaligned shard that starts on a
                        # batch's last ray): the library needs B >= 2 like the reference (nerf.py:208 .squeeze()), so the piece is
                        # launched with its last ray repeated and the copy cropped -- rays are independent and ray0_near_far is handed
                        # over explicitly, so the real ray's bits do not change
                        k = MIN_CALL_RAYS - (e - s)
                        r_, c_, p_ = (torch.cat((x, x[-1:].expand(k, *x.shape[1:]))) for x in (r_, c_, p_))
                    c, f = self._launch(ps, r_, c_, p_, K_inv)
                    C_c[s - lo:e - lo] = c[: e - s]
                    C_f[s - lo:e - lo] = f[: e - s]
        finally:
            self.ray0_near_far, self._ws_capacity = prev_ray0, prev_cap
        if getattr(self, "bf16_mlp", False):
            self.read_status()  # a frame is not handed out on a poisoned weight image (raises on STATUS_PREP_TIMEOUT; one sync per frame)
        return C_c, C_f


def fuse_plan(near_far0, n: int, batch: int, lo: int = 0, hi: int | None = None, fuse_rays: int = 16384):
    """Kernel calls that render rays [lo, hi) of an n-ray list with the reference's batch semantics.  The reference renders the list in
    batches [g*batch, (g+1)*batch) and its resampler takes the coarse spacing from ray 0 OF EACH BATCH (nerf.py:233, quirk Q6) -- the
    only cross-ray term of the path.  Consecutive batches whose ray 0 has the same (near, far) can therefore share one call that is
    handed that pair: same bits per ray, fewer and fuller launches.  near_far0: [ceil(n / batch)][2] (near, far) of every batch's ray 0
    (host floats).  Returns [(s, e, near, far)], s < e, each call inside one run of equal pairs and at most `fuse_rays` rays long
    (at least one batch), pieces on the batch grid wherever the range allows it."""
    hi = n if hi is None else hi
    per_call = max(batch, fuse_rays // batch * batch)
    plan = []
    s = lo
    while s < hi:
        g = s // batch
        nf = (float(near_far0[g][0]), float(near_far0[g][1]))
        e = min((g + 1) * batch, hi)
        while e < hi and e - s + batch <= per_call:  # take the next batch along if its ray 0 agrees
            g2 = e // batch
            if (float(near_far0[g2][0]), float(near_far0[g2][1])) != nf:
                break
            e = min((g2 + 1) * batch, hi)
        plan.append((s, e, nf[0], nf[1]))
        s = e
    return plan


class _RayLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C_c, C_f, C_t):
        C_c, C_f, C_t = C_c.contiguous(), C_f.contiguous(), C_t.contiguous().to(C_c.device, torch.float32)
        B = C_c.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=C_c.device)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dCc = torch.empty_like(C_c) if need else None
        dCf = torch.empty_like(C_f) if need else None
        _abi.check(_abi.lib().nerf_hip_ray_loss(C_c.data_ptr(), C_f.data_ptr(), C_t.data_ptr(), B, loss.data_ptr(),
                                                dCc.data_ptr() if need else None, dCf.data_ptr() if need else None,
                                                torch.cuda.current_stream(C_c.device).cuda_stream))
        if need:
            ctx.save_for_backward(dCc, dCf)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dCc, dCf = ctx.saved_tensors
        return g * dCc, g * dCf, None


def __getattr__(name):  # `from nerf import NeRFRunner` (main.py:4) without a circular import at module load
    if name == "NeRFRunner":
        from .train import NeRFRunner

        return NeRFRunner
    raise AttributeError(name)
