# -*- coding: utf-8 -*-
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub(d, prefix):
    """Entries of a flat golden dict under `prefix/`, with the prefix stripped."""
    p = prefix + "/"
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


def rel_err(a, b):
    """Norm-wise relative error max|a-b| / max|b| (the 1e-5 criterion of BASELINE.json's north_star is applied to this:
    element-wise ratios are meaningless for gradient entries that cancel to ~0)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    den = np.abs(b).max()
    num = np.abs(a - b).max()
    return float(num / den) if den > 0 else float(num)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def tune():
    """tune(GEMM16_TILE=256, ...): set tuning knobs of libtrs_hip.so for the test (trs_tuning_set; the library reads the
    TRS_* environment only once, at its first use), defaults restored afterwards."""
    from torchrecsys_amd import _lib
    touched = []

    def _set(**knobs):
        lib = _lib.load()
        for k, v in knobs.items():
            _lib.check(lib.trs_tuning_set(k.encode(), int(v), 0), "trs_tuning_set")
            touched.append(k)
    yield _set
    lib = _lib.load()
    for k in touched:
        lib.trs_tuning_set(k.encode(), 0, 1)
