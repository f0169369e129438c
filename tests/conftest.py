import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import nerf_oracle

    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))  # the fixtures were generated with 8 threads
    return nerf_oracle


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_inputs(g):
    t = torch.from_numpy
    return (t(g["row"]), t(g["col"]), t(g["poses_bound"]), t(g["K_inv"]), t(g["C_true"]))


def max_rel(a, b):
    """max |a-b| / max|b|  (the 'max-rel' of SURVEY 8d)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elementwise_rel(a, b, floor=1e-6):
    """max over elements of |a-b| / max(|b|, floor): the element-wise reading of 'rel' (north_star: 1e-4 rel fp32)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())


def l2_rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def pkg():
    import nerf_tiny_amd

    return nerf_tiny_amd
