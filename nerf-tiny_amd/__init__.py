"""MI355X-native NeRF volume-rendering hot path (drop-in for D-Hank/NeRF-tiny's NeRFModel).

Layout:  csrc/ (HIP kernels + C ABI -> libnerf_hip.so), _abi.py (ctypes binding),
nerf.py (host-side mirror of the reference's ``nerf`` module surface), ops.py (stage-level calls),
parallel.py (ray-batch data parallelism), data.py / train.py / main.py (the callers: sampler, optimizer, runner).
"""
from . import _abi, data, ops, parallel, train  # noqa: F401
from .nerf import NeRFModel, Network, Encoder  # noqa: F401
from .train import FusedAdam, NeRFRunner  # noqa: F401
HAS_BACKWARD = True
