import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference); skipped when absent")


@pytest.fixture(scope="session")
def orc():
    from tests import _orc
    return _orc.oracle()


@pytest.fixture(scope="session")
def ref():
    from tests import _orc
    r = _orc.reference()
    if r is None:
        pytest.skip("oracle/_ref/libhevcref.so not built (reference sources absent)")
    return r


@pytest.fixture(scope="session", autouse=True)
def _torch_brings_the_gpu_up_first():
    """On the GPU box: let torch initialise HIP before the library does.  The other order (a test file that only uses the
    C ABI first, then a torch-based one in the same process) makes torch report 'No HIP GPUs are available' on this image."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
            torch.cuda.synchronize()
    except Exception:       # noqa: BLE001 -- no GPU, or no torch: the CPU suite does not need either
        pass
    yield
