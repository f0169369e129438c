"""``import loader`` -- the reference's data module name (/root/reference/nerf.py:21, loader.py): an alias of
``nerf_tiny_amd.data`` (``NeRFDataset``, ``create_npy``, ``convert_npy``, ``data_preprocess``, NEAR/FAR factors)."""
import sys as _sys

import nerf_tiny_amd.data as _impl

_sys.modules[__name__] = _impl
