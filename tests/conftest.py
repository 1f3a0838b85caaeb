import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from deepdenoiser_amd import _lib
    return _lib.load()


def pytest_sessionfinish(session, exitstatus):
    """Every oracle comparison of a -m gpu session with its measured error and its gate -> gpurun_out/parity_errors.txt (copied to profiles/)."""
    try:
        import gpu_util
    except Exception:
        return
    if not gpu_util.RECORDS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, os.environ.get("DD_PARITY_FILE", "parity_errors.txt")), "w") as f:
        f.write("# test | compared tensor | measured rel-L2 (or scalar) | gate | measured/gate\n")
        for test, name, e, tol in gpu_util.RECORDS:
            f.write("%s | %s | %.3e | %.1e | %.2f%s\n" % (test, name, e, tol, e / tol if tol else 0.0, "  <-- OVER" if e > tol else ""))
