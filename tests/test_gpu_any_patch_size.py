"""GPU tests of the any-patch-size table kernel (round 5, k_bm_scan_any): for the patch sizes that also have dedicated kernels it
must write the same bits -- raw disparity tables, self-search scores, selections, sums -- as the first-generation kernel whose
layout it shares (LFBM5D_SCAN_V1); other patch sizes are compared with the oracle in test_gpu_parity.py (PASS_CASES ht-k10-*, ...)."""
import numpy as np
import pytest

from test_gpu_parity import gpu_pass, window

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lfbm5d_amd as L
    c = L.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", [(1, (8, 8, 3, 16, 4, "id", "sadct", "haar"), 96), (2, (8, 6, 2, 8, 3, "dct", "sadct", "haar"), (72, 150)),
                                  (1, (4, 6, 2, 12, 4, "dct", "sadct", "haar"), 72)], ids=["k16", "k8-ragged", "k12"])
def test_plain_table_kernel_equals_the_dedicated_ones(ctx, monkeypatch, case):
    step, pk, crop = case
    win, Wb, Hb, Cc = window(25.0, pk, crop)
    basic = np.ascontiguousarray(0.5 * win + 0.5 * np.roll(win, 1, axis=1)) if step == 2 else None
    out = {}
    for env in ("LFBM5D_SCAN_V1", "LFBM5D_SCAN_ANY"):
        monkeypatch.delenv("LFBM5D_SCAN_V1", raising=False)
        monkeypatch.delenv("LFBM5D_SCAN_ANY", raising=False)
        monkeypatch.setenv(env, "1")
        num, den = gpu_pass(ctx, step, 25.0, pk, win, basic, Wb, Hb, Cc)
        assert ctx.last_scan_version() == 1
        refs, idx, cnt, best, shape = ctx.last_bm(pk[0], 9, Wb * Hb)
        out[env] = (num, den, idx, cnt, best, shape, ctx.last_tables(), ctx.last_scores())
    a, b = out["LFBM5D_SCAN_V1"], out["LFBM5D_SCAN_ANY"]
    k, nDisp = pk[3], pk[2]
    # raw disparity tables: every entry either kernel writes (the skewed layout's corners hold nothing)
    ta, tb = a[6], b[6]
    assert ta.shape == tb.shape
    written = ta != tb
    assert written.mean() < 0.25                                   # (unwritten corners may differ: stale memory)
    regr, regc = slice(nDisp, Hb - k - nDisp + 1), slice(nDisp, Wb - k - nDisp + 1)
    for st in (0, 1, 2, 3, 5, 6, 7, 8):
        assert np.array_equal(a[4][st].reshape(Hb, Wb)[regr, regc], b[4][st].reshape(Hb, Wb)[regr, regc])
        assert np.array_equal(a[5][st].reshape(Hb, Wb)[regr, regc], b[5][st].reshape(Hb, Wb)[regr, regc])
    assert np.array_equal(a[7], b[7])                              # self-search scores, entry for entry
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
