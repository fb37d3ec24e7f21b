"""Shared fixtures.  `-m "not gpu"`: oracle vs golden / fp64, host logic, ABI exports (no GPU needed).
`-m gpu`: the parity tests proper — HIP path through the C ABI vs the CPU restatement, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver at round end / via gpurun)")


@pytest.fixture(scope="session")
def pkg():
    """The product package.  The shared library is git-ignored, so a fresh checkout builds it first (hipcc
    cross-compiles for gfx950 without a GPU; on the GPU box the prebuilt .so travels with the snapshot)."""
    p = ge.load_package()
    if not os.path.exists(p.api.LIB_PATH):
        ge.build()
    return p


@pytest.fixture(scope="session")
def O():
    return ge.load_oracle()


@pytest.fixture(scope="session")
def reg(pkg):
    """One GPU context for the whole session.  Fails loudly without the HIP library/device: no fallback."""
    r = pkg.Registrar(0)
    yield r
    r.close()


def nan_equal_bits(a: np.ndarray, b: np.ndarray) -> bool:
    """Bitwise equality of fp32 arrays, except that any NaN matches any NaN (x86 and gfx950 produce
    different default-NaN sign bits)."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(both_nan | (a.view(np.uint32) == b.view(np.uint32))))
