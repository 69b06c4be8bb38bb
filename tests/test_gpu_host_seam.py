"""GPU tests of the host seam (round 5): lfbm5d_*_host / *_host_sai and the C++ drop-in stream the caller's SAIs through the
window graph -- a SAI goes up when the first window that needs it is enqueued, its outputs come down behind the last window on
it -- instead of copying four light fields around the job.  The bar is bit-identity with the device-resident entry points for
every light field a call returns (LF_noisy after its lossy colour round trips, the basic estimate, the result), including the
sequential redo of a job whose graph turned out incomplete (LFBM5D_FORCE_REDO)."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu

ENV = ("LFBM5D_EMULATE_WORLD", "LFBM5D_DATA_DRIVEN_SCHEDULE", "LFBM5D_STEP_SHARDING", "LFBM5D_LANES", "LFBM5D_MAX_WINDOWS", "LFBM5D_FUSED",
       "LFBM5D_HOST_BLOCKING", "LFBM5D_FORCE_REDO")

HT = (4, 6, 2, 8, 4, "id", "sadct", "haar")
WIEN = (8, 6, 2, 8, 4, "dct", "sadct", "haar")


@pytest.fixture()
def ctx():
    import lfbm5d_amd as L
    c = L.Context(0)          # a fresh context per test: buffer (re)allocation paths are part of what is tested
    yield c
    c.close()


def _clean_env(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def _device(ctx, kind, P1, P2, noisy, basic_in, mask, aw, ah, W, H, mj, an=(1, 1)):
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic = torch.from_numpy(basic_in).cuda() if basic_in is not None else torch.zeros_like(d_noisy)
    d_den = torch.zeros_like(d_noisy)
    if kind == 1:
        ctx.step1(P1, d_noisy, mask, d_basic, mj, aw, ah, an[0], W, H, 3)
    elif kind == 2:
        ctx.step2(P2, d_noisy, mask, d_basic, d_den, mj, aw, ah, an[1], W, H, 3)
    else:
        ctx.denoise(P1, P2, d_noisy, mask, d_basic, d_den, mj, aw, ah, an[0], an[1], W, H, 3)
    return d_noisy.cpu().numpy(), d_basic.cpu().numpy(), d_den.cpu().numpy(), ctx.last_windows()


def _host(ctx, kind, P1, P2, noisy, basic_in, mask, aw, ah, W, H, mj, an=(1, 1), per_sai=False):
    n = noisy.copy()
    b = basic_in.copy() if basic_in is not None else np.full_like(noisy, -7.0)
    d = np.full_like(noisy, -7.0)
    if per_sai:   # one array per SAI, like the reference's vector<vector<float>>; empty SAIs hold nothing
        ln = [n[i] if mask[i] else None for i in range(len(mask))]
        lb = [b[i] if mask[i] else None for i in range(len(mask))]
        ld = [d[i] if mask[i] else None for i in range(len(mask))]
        args = (ln, lb, ld)
    else:
        args = (n, b, d)
    if kind == 1:
        ctx.step1(P1, args[0], mask, args[1], mj, aw, ah, an[0], W, H, 3)
    elif kind == 2:
        ctx.step2(P2, args[0], mask, args[1], args[2], mj, aw, ah, an[1], W, H, 3)
    else:
        ctx.denoise(P1, P2, args[0], mask, args[1], args[2], mj, aw, ah, an[0], an[1], W, H, 3)
    return n, b, d, ctx.last_windows()


def _same(a, b, mask, kind):
    """the light fields of the non-empty SAIs (what either form defines); step 1 has no denoised output"""
    m = mask != 0
    ok = np.array_equal(a[0][m], b[0][m]) and np.array_equal(a[1][m], b[1][m]) and np.array_equal(a[3], b[3])
    return ok and (kind == 1 or np.array_equal(a[2][m], b[2][m]))


CASES = [
    # name, ah, aw, H, W, holes, colour space, major, lanes
    ("7x9-opp", 7, 9, 64, 64, (), "opp", "row", "2"),
    ("7x6-holes-col-yuv", 7, 6, 56, 64, (0, 11, 40), "yuv", "col", "3"),
    ("5x5-rgb-one-lane", 5, 5, 64, 72, (), "rgb", "row", "1"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_streamed_host_seam_is_bit_identical_to_device_buffers(ctx, monkeypatch, case):
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    name, ah, aw, Hs, Ws, holes, cs, major, lanes = case
    mj = L.ROWMAJOR if major == "row" else L.COLMAJOR
    _, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, Hs, Ws), 25.0)
    mask = np.ones(ah * aw, np.uint32)
    mask[list(holes)] = 0
    P1 = core.make_params(25.0, 2.7, *HT, color_space=cs)
    P2 = core.make_params(25.0, 2.7, *WIEN, color_space=cs)
    _clean_env(monkeypatch)
    monkeypatch.setenv("LFBM5D_LANES", lanes)
    ref1 = _device(ctx, 1, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, mj)
    basic_in = ref1[1]
    ref2 = _device(ctx, 2, P1, P2, ref1[0], basic_in, mask, aw, ah, Ws, Hs, mj)
    ref3 = _device(ctx, 3, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, mj)
    assert np.array_equal(ref3[2], ref2[2])                       # (the job equals the two calls: test_gpu_denoise.py)
    for per_sai in (False, True):
        for blocking in (False, True):
            if blocking:
                monkeypatch.setenv("LFBM5D_HOST_BLOCKING", "1")
            else:
                monkeypatch.delenv("LFBM5D_HOST_BLOCKING", raising=False)
            h1 = _host(ctx, 1, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, mj, per_sai=per_sai)
            assert _same(h1, ref1, mask, 1), (name, "step 1", per_sai, blocking)
            h2 = _host(ctx, 2, P1, P2, ref1[0], basic_in, mask, aw, ah, Ws, Hs, mj, per_sai=per_sai)
            assert _same(h2, ref2, mask, 2), (name, "step 2", per_sai, blocking)
            h3 = _host(ctx, 3, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, mj, per_sai=per_sai)
            assert _same(h3, ref3, mask, 3), (name, "job", per_sai, blocking)
            if not per_sai:   # the flat form returns zeros for the outputs of empty SAIs and leaves their input alone
                e = mask == 0
                assert not h3[1][e].any() and not h3[2][e].any() and np.array_equal(h3[0][e], noisy[e])


def test_streamed_host_seam_with_a_window_limit(ctx, monkeypatch):
    """LFBM5D_MAX_WINDOWS leaves SAIs no window touches -- in one step or in both: they travel too and keep the step's input."""
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    ah, aw, Hs, Ws = 7, 9, 64, 64
    _, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, Hs, Ws), 25.0)
    mask = np.ones(ah * aw, np.uint32)
    P1, P2 = core.make_params(25.0, 2.7, *HT), core.make_params(25.0, 2.7, *WIEN)
    _clean_env(monkeypatch)
    monkeypatch.setenv("LFBM5D_MAX_WINDOWS", "4")
    for kind in (1, 3):
        ref = _device(ctx, kind, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, L.ROWMAJOR)
        got = _host(ctx, kind, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, L.ROWMAJOR)
        assert _same(got, ref, mask, kind), kind


def test_redo_of_an_incomplete_graph_on_a_fresh_context(monkeypatch):
    """The graph form assumes one pass per window and checks the coverage counts at the end; a job that violates it is redone
    window after window (LFBM5D_FORCE_REDO plays that).  On a FRESH context the redo used to go through a freed buffer (round-4
    advisor finding): every variant here starts from a new context.  The redo of the streamed host form starts from the light
    field as it arrived, whatever the streamed outputs have overwritten in the caller's buffers."""
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    ah, aw, Hs, Ws = 5, 6, 56, 64
    _, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, Hs, Ws), 25.0)
    mask = np.ones(ah * aw, np.uint32)
    P1, P2 = core.make_params(25.0, 2.7, *HT), core.make_params(25.0, 2.7, *WIEN)
    _clean_env(monkeypatch)
    monkeypatch.setenv("LFBM5D_LANES", "1")
    c0 = L.Context(0)
    ref1 = _device(c0, 1, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, L.ROWMAJOR)
    ref2 = _device(c0, 2, P1, P2, ref1[0], ref1[1], mask, aw, ah, Ws, Hs, L.ROWMAJOR)
    c0.close()
    monkeypatch.setenv("LFBM5D_LANES", "2")
    monkeypatch.setenv("LFBM5D_FORCE_REDO", "1")
    for kind, ref, n_in, b_in in ((1, ref1, noisy, None), (2, ref2, ref1[0], ref1[1]), (3, ref2, noisy, None)):
        for form in (_device, _host):
            c = L.Context(0)
            got = form(c, kind, P1, P2, n_in, b_in, mask, aw, ah, Ws, Hs, L.ROWMAJOR)
            c.close()
            if kind == 3:   # the job's windows: both steps'
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), (kind, form.__name__)
            else:
                assert _same(got, ref, mask, kind), (kind, form.__name__)


def test_cpp_dropin_on_vectors_equals_the_device_form(ctx, monkeypatch):
    """run_bm5d_1st_step + run_bm5d_2nd_step of liblfbm5d_dropin.so (the reference's signatures, src/bm5d.h:11-62) on
    vector<vector<float>> light fields -- the vectors' own storage goes to the library, nothing is flattened -- and run_bm5d."""
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    ah, aw, Hs, Ws = 5, 7, 64, 56
    _, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, Hs, Ws), 25.0)
    mask = np.ones(ah * aw, np.uint32)
    mask[[3, 30]] = 0
    P1, P2 = core.make_params(25.0, 2.7, *HT), core.make_params(25.0, 2.7, *WIEN)
    _clean_env(monkeypatch)
    ref = _device(ctx, 3, P1, P2, noisy, None, mask, aw, ah, Ws, Hs, L.ROWMAJOR)
    m = mask != 0
    for one_job in (False, True):
        ms, n, b, d = core.dropin_probe(noisy, mask, aw, ah, Ws, Hs, 3, 25.0, 2.7, HT, WIEN, one_job=one_job, reps=2)
        assert ms.shape == (2, 2) and (ms[:, 0] > 0).all()
        assert np.array_equal(n[m], ref[0][m]) and np.array_equal(b[m], ref[1][m]) and np.array_equal(d[m], ref[2][m]), one_job
