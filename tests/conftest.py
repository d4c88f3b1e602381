import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this process")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _progress_note(request):
    """One line per test into gpurun_out/pytest_progress.log: `pytest -q` prints nothing while a long test runs,
    and the GPU box takes three silent minutes for a hang."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "pytest_progress.log"), "a") as f:
            f.write(request.node.nodeid + "\n")
    except OSError:
        pass
    yield
