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
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def relerr(a, b):
    """Norm-wise relative error ||a-b|| / ||b|| in float64 (SURVEY hard part 1)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


# north_star tolerance: 1e-4 relative, fp32
RTOL = 1e-4


def assert_close(a, b, rtol=RTOL, atol_scale=None, what=""):
    """Norm-wise rtol plus a max-abs bound of atol_scale*max|b| (default 10*rtol)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    r = relerr(a, b)
    assert r <= rtol, "%s norm-wise rel err %.3e > %.1e" % (what, r, rtol)
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    bound = (atol_scale if atol_scale is not None else 10 * rtol) * max(scale, 1e-30)
    m = maxabs(a, b)
    assert m <= bound, "%s max abs err %.3e > %.3e" % (what, m, bound)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.build()
    return oracle


def assert_close_knife_edge(a, b, rtol=RTOL, max_outlier_frac=2e-3, what=""):
    """For piecewise-constant quantities (bilinear-sample gradients w.r.t. the sampling position): a sample whose
    position sits within fp32 rounding of a texel boundary may legitimately pick the neighbouring cell, which
    changes its gradient by O(1).  Allow a tiny fraction of such samples, require the rest to match norm-wise."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    assert a.shape == b.shape
    scale = np.abs(b).max()
    bad = np.abs(a - b) > 1e-3 * scale
    frac = bad.mean()
    assert frac <= max_outlier_frac, "%s: %.3e of the samples differ (allowed %.1e)" % (what, frac, max_outlier_frac)
    r = np.linalg.norm(a[~bad] - b[~bad]) / np.linalg.norm(b[~bad])
    assert r <= rtol, "%s norm-wise rel err %.3e > %.1e (outliers excluded: %.2e)" % (what, r, rtol, frac)
