"""``import nerf`` -- the module name the reference's driver uses (``from nerf import NeRFRunner``, /root/reference/main.py:4).

This file makes ``nerf`` BE ``nerf_tiny_amd.nerf`` (same module object: ``nerf.device``, ``nerf.NeRFModel``,
``nerf.NeRFRunner`` ...), so the reference's main.py:36-56 call sequence runs against the MI355X path when this
repository's root is on ``sys.path`` ahead of the reference's directory.
"""
import sys as _sys

import nerf_tiny_amd.nerf as _impl
from nerf_tiny_amd.train import FusedAdam, NeRFRunner  # noqa: F401

_impl.NeRFRunner = NeRFRunner
_impl.FusedAdam = FusedAdam
_sys.modules[__name__] = _impl
