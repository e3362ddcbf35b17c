import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import ninpol_oracle
    ninpol_oracle.build_port()
    return ninpol_oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first(request):
    """PyTorch's wheel carries its own HIP / HSA runtime (torch/lib/libamdhip64.so) next to the system one libninpol_amd.so links:
    two runtimes in one process.  The second one to open the device can fail ("No HIP GPUs are available") once the first holds tens
    of GB of mappings -- tests/test_gpu_scale.py run on its own met that at its last test.  On a GPU box torch therefore opens the
    device before any test does (GPU selections only: the CPU suite must not touch it)."""
    expr = request.config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:   # no torch, no device: the tests that need one say so themselves
            pass
    yield
