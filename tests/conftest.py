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
