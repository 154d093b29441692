import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip or pass vacuously: the HIP path is the product.
    expr = config.getoption("-m") or ""
    wants_gpu = "gpu" in expr and "not gpu" not in expr
    if wants_gpu and any(item.get_closest_marker("gpu") for item in items) and not _has_gpu():
        raise pytest.UsageError("-m gpu was selected but no HIP device is visible: these tests drive libqn_hip.so on an MI355X "
                                "and there is no CPU fallback to test instead")


@pytest.fixture(scope="session")
def qo():
    from oracle import qn_oracle
    qn_oracle.build()
    return qn_oracle


@pytest.fixture(scope="session")
def qn():
    """The product package (optimization-solvers_amd/), loaded under an importable name."""
    import __graft_entry__ as ge
    return ge.load_package()
