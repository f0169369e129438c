"""CPU: the C-ABI library loads and exports every symbol include/nerf_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_hip_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(pkg):
    assert _declared() == sorted(pkg._abi.EXPORTS)


def test_library_exports_all_symbols(pkg):
    lib = ctypes.CDLL(pkg._abi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert pkg._abi.lib().nerf_hip_abi_version() == pkg._abi.NERF_HIP_ABI_VERSION


def test_ws_bytes_and_errors(pkg):
    n0 = pkg._abi.ws_bytes(4096, 64, 128, 0)
    n1 = pkg._abi.ws_bytes(4096, 64, 128, pkg._abi.SAVE_FOR_BACKWARD)
    assert 0 < n0 < n1
    with pytest.raises(pkg._abi.NerfHipError):
        pkg._abi.ws_bytes(1, 64, 128, 0)  # B = 1 is unsupported like the reference (quirk Q7)
    with pytest.raises(pkg._abi.NerfHipError):
        pkg._abi.ws_bytes(8, 1, 128, 0)
    # the kernels index samples with 32-bit integers: a batch of >= 2^31 samples is refused, not wrapped
    with pytest.raises(pkg._abi.NerfHipError, match="2\\^31"):
        pkg._abi.ws_bytes(2_000_000, 1024, 1024, pkg._abi.SAVE_FOR_BACKWARD)
    assert pkg._abi.ws_bytes(1_000_000, 64, 128, 0) > 0
    # the opt-in split-fp32 inference mode brings its own packed image (2.1 MB) and nothing else; with the bf16 flag it is ignored
    d = pkg._abi.ws_bytes(4096, 64, 128, pkg._abi.SPLIT_MLP) - n0
    assert 2_100_000 < d < 2_300_000
    assert pkg._abi.ws_bytes(4096, 64, 128, pkg._abi.SPLIT_MLP | pkg._abi.BF16_MLP) == pkg._abi.ws_bytes(4096, 64, 128, pkg._abi.BF16_MLP)


def test_no_cpu_fallback(pkg):
    import torch

    m = pkg.NeRFModel(64, 128, 4)
    row = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        m(row, row, torch.zeros(4, 17, dtype=torch.float64), torch.eye(3))
    with pytest.raises(RuntimeError):
        m.network(None)


def test_product_does_not_import_oracle():
    pk = os.path.join(ROOT, "nerf-tiny_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"import\s+nerf_oracle|from\s+nerf_oracle|oracle/", text), f
