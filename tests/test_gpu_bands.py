"""GPU tests of the banded pass (round 5): group kernel + aggregation launched band after band of reference rows (LFBM5D_BAND_MB; taken
automatically when the filtered-patch buffer of a pass would be too large) must leave exactly the sums of the single launch -- the
aggregation adds up in raster order of the reference patches (core:484-528), and bands are cut along that order."""
import numpy as np
import pytest

import helpers as Hh
from test_gpu_parity import gpu_pass, window

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lfbm5d_amd as L
    c = L.Context(0)
    yield c
    c.close()


CASES = [
    # name, step, sigma, params, crop, grey
    ("ht-k16-n8", 1, 25.0, (8, 8, 3, 16, 4, "id", "sadct", "haar"), 96, False),
    ("ht-k16-bior-n1", 1, 50.0, (1, 6, 2, 16, 3, "bior", "sadct", "haar"), 96, False),
    ("wien-dct-n16", 2, 25.0, (16, 8, 3, 8, 4, "dct", "sadct", "haar"), 96, False),
    ("ht-k12-generic", 1, 25.0, (4, 6, 2, 12, 4, "dct", "sadct", "haar"), 72, False),
    ("wien-wide-p3", 2, 25.0, (8, 9, 3, 8, 3, "dct", "sadct", "haar"), (72, 150), False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_banded_pass_is_bit_identical_to_the_single_launch(ctx, monkeypatch, case):
    name, step, sigma, pk, crop, grey = case
    win, Wb, Hb, Cc = window(sigma, pk, crop, grey)
    basic = np.ascontiguousarray(0.5 * win + 0.5 * np.roll(win, 1, axis=1)) if step == 2 else None
    rng = np.random.default_rng(3)
    num0 = (rng.random(win.shape) * 4).astype(np.float32)    # sums a previous window left behind: the bands add on top, in order
    den0 = (rng.random(win.shape) * 0.01).astype(np.float32)
    monkeypatch.delenv("LFBM5D_BAND_MB", raising=False)
    n_ref, d_ref = gpu_pass(ctx, step, sigma, pk, win, basic, Wb, Hb, Cc, num=num0.copy(), den=den0.copy())
    launches = []
    for mb in ("1", "3", "17"):
        monkeypatch.setenv("LFBM5D_BAND_MB", mb)
        ctx.reset_stats()
        n_b, d_b = gpu_pass(ctx, step, sigma, pk, win, basic, Wb, Hb, Cc, num=num0.copy(), den=den0.copy())
        launches.append(ctx.stats().launches_group)
        assert np.array_equal(n_b, n_ref) and np.array_equal(d_b, d_ref), (name, mb)
    assert launches[0] > 1, "a 1 MB band must cut these passes into several launches"


def test_banded_subset_pass(ctx, monkeypatch):
    """greyscale light fields: the subset pass (pst != cst) works on a LIST of reference patches -- bands are slices of it."""
    sigma, pk = 25.0, (4, 6, 2, 8, 3, "id", "sadct", "haar")
    win, Wb, Hb, Cc = window(sigma, pk, 64, grey=True)
    monkeypatch.delenv("LFBM5D_BAND_MB", raising=False)
    n0, d0 = gpu_pass(ctx, 1, sigma, pk, win, None, Wb, Hb, Cc)                       # centre pass
    d0[:, : d0.shape[1] // 3] = 0                                                     # leave a third of every SAI uncovered
    n0[:, : n0.shape[1] // 3] = 0
    proc = np.zeros(9, np.uint32); proc[4] = 1
    ref = gpu_pass(ctx, 1, sigma, pk, win, None, Wb, Hb, Cc, num=n0.copy(), den=d0.copy(), proc=proc, cst=4, pst=1)
    monkeypatch.setenv("LFBM5D_BAND_MB", "1")
    got = gpu_pass(ctx, 1, sigma, pk, win, None, Wb, Hb, Cc, num=n0.copy(), den=d0.copy(), proc=proc, cst=4, pst=1)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert not np.array_equal(ref[1], d0)
