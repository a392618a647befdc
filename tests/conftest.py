import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _restore_mlp_precision():
    """tests switch the MFMA operand mode (mlp.set_precision); whatever a test leaves behind, the next one starts from the
    mode this process was started with (the library default bf16x3, or NUDF_PRECISION)."""
    try:
        from neuraludf_amd import mlp
    except Exception:          # pragma: no cover - the package always imports (no GPU needed for that)
        yield
        return
    old, old_fwd, old_bwd = mlp.PRECISION, mlp.FWD_F16X2, mlp.BWD_F16X2
    yield
    mlp.PRECISION, mlp.FWD_F16X2, mlp.BWD_F16X2 = old, old_fwd, old_bwd
