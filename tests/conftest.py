"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on CPU only."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_unstructured_square():
    """The reference's own mesh fixture (meshes/unstructured_square, read like tests/load_unstructured_square.cpp:20-54)."""
    d = GOLDEN / "unstructured_square"
    n_pts, n_elem = (int(v) for v in (d / "info.txt").read_text().split())
    xy = np.loadtxt(d / "coordinates.txt").reshape(n_pts, 2)
    elems = np.loadtxt(d / "elements.txt", dtype=np.int64).reshape(n_elem, 4)
    return xy, elems


@pytest.fixture(scope="session")
def unstructured_square():
    return load_unstructured_square()


@pytest.fixture(scope="session")
def cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the product has no CPU fallback)")
    import cuddhelmholtz_amd as cd

    cd.use_torch_stream()
    return torch.device("cuda:0")
