"""ctypes binding of libnerf_hip.so (include/nerf_hip.h).

The library is the product: if it is missing or does not export a declared symbol this module
raises -- there is no CPU or eager-PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_HIP_LIB", os.path.join(_HERE, "libnerf_hip.so"))  # override: diagnostic builds only

NERF_HIP_ABI_VERSION = 6
SAVE_FOR_BACKWARD = 1 << 0
FORCE_TILE_KERNEL = 1 << 1
BF16_MLP = 1 << 2
WEIGHTS_UNCHANGED = 1 << 3
SPLIT_MLP = 1 << 4
CORRECTED = 1 << 5
STATUS_RESAMPLE_INDEX = 1 << 0
STATUS_PREP_TIMEOUT = 1 << 1

_p = C.c_void_p
_PROTOS = {
    "nerf_hip_abi_version": (C.c_int, []),
    "nerf_hip_last_error": (C.c_char_p, []),
    "nerf_hip_ws_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "nerf_hip_ws_offset": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]),
    "nerf_hip_forward": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, _p, C.c_size_t, C.c_int, _p]),
    "nerf_hip_backward": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, C.c_size_t, C.c_int, _p]),
    "nerf_hip_backward_overlap": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, C.c_size_t, C.c_int, _p, _p]),
    "nerf_hip_ray_loss": (C.c_int, [_p, _p, _p, C.c_int, _p, _p, _p, _p]),
    "nerf_hip_train_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, _p, _p, _p, C.c_size_t, C.c_int, _p, _p]),
    "nerf_hip_read_status": (C.c_int, [_p, C.c_size_t, C.POINTER(C.c_uint32), _p]),
    "nerf_hip_read_status_sticky": (C.c_int, [_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_int, _p]),
    "nerf_hip_profile_begin": (C.c_int, [C.c_int]),
    "nerf_hip_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    "nerf_hip_adam_step": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _p]),
    "nerf_hip_gather_rays": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p]),
    "nerf_hip_rays": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, _p, _p, _p, _p]),
    "nerf_hip_field": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p, _p, _p, _p, _p, C.c_size_t, _p]),
    "nerf_hip_field_bf16": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p, _p, _p, C.c_size_t, _p]),
    "nerf_hip_coarse_composite": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p]),
    "nerf_hip_coarse_composite_backward": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p]),
    "nerf_hip_merge_composite": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, _p, _p]),
}
EXPORTS = tuple(_PROTOS)

_lib = None


class NerfHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Loads libnerf_hip.so once.  Raises if it is absent (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfHipError(f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}`; "
                               "this package has no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L.nerf_hip_abi_version() != NERF_HIP_ABI_VERSION:
            raise NerfHipError("libnerf_hip.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise NerfHipError(f"libnerf_hip error {rc}: {lib().nerf_hip_last_error().decode()}")


def ws_bytes(B: int, Nc: int, Nf: int, flags: int) -> int:
    n = C.c_size_t(0)
    check(lib().nerf_hip_ws_bytes(B, Nc, Nf, flags, C.byref(n)))
    return int(n.value)


KERNEL_NAMES = ("pack_weights", "rays", "field_fwd_coarse", "coarse_composite", "field_fwd_fine", "merge_composite",
                "bwd_merge", "bwd_field_fine", "bwd_coarse", "bwd_field_coarse", "bwd_dw", "render_pair")


def profile_begin(max_launches: int) -> None:
    check(lib().nerf_hip_profile_begin(max_launches))


def profile_end() -> dict:
    """-> {kernel name: (total ms, launches)}"""
    n = len(KERNEL_NAMES)
    ms = (C.c_double * n)()
    cnt = (C.c_int * n)()
    check(lib().nerf_hip_profile_end(ms, cnt, n))
    return {KERNEL_NAMES[i]: (ms[i], cnt[i]) for i in range(n) if cnt[i]}


def ws_view(ws, B, Nc, Nf, flags, name, shape, dtype=None):
    """A typed view of a named workspace buffer (tests / debugging)."""
    import torch

    off = C.c_size_t(0)
    check(lib().nerf_hip_ws_offset(B, Nc, Nf, flags, name.encode(), C.byref(off)))
    dtype = dtype or torch.float32
    n = 1
    for d in shape:
        n *= d
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    return ws[off.value: off.value + nbytes].view(dtype).view(*shape)


def ptr_array(tensors) -> "C.Array":
    """HOST array of device pointers (weights24 / dweights24)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def f32_array(values) -> "C.Array":
    return (C.c_float * len(values))(*[float(v) for v in values])
