"""Independent cross-checks of the oracle's UNPINNED parts (block matching, SADCT, slab filters) against short
numpy models written from the published definitions, not from the oracle's code:

  * block matching: brute-force sums of squared differences in float64 (no integral image, no recurrence), with the
    reference's documented construction quirks stated as definitions (zero band, mirrored test, 2*threshold entries,
    scan order; core:3301-3611);
  * shape-adaptive DCT: per-row / per-column orthonormal DCT-II of the in-shape entries with scipy.fft.dct, compaction to
    the row / column start, the 0.5/sqrt(2) factor (core:1969-2116), on all 511 masks WITH VALUES;
  * hard-threshold / Wiener slab filters: orthonormal Haar matrices, threshold lambda*sigma*sqrt(2), e^2/(e^2+sigma^2)
    (core:2408-2505, :2826-2925).

None of this turns the oracle into a pinned one (only reference-held vectors or the reference compiled without
stand-ins can: DESIGN.md section 5); it removes the risk of a restatement error in these routines.
"""
import numpy as np
import pytest
import scipy.fft

import helpers as Hh
from oracle import oracle as O


# ------------------------------------------------------------------------------------------------------------------
# block matching
# ------------------------------------------------------------------------------------------------------------------
def _box(D, k):
    """S[i][j] = sum of D[i:i+k, j:j+k] (float64), same shape as D with zeros where the box leaves the array."""
    H, W = D.shape
    c = np.zeros((H + 1, W + 1))
    c[1:, 1:] = D.cumsum(0).cumsum(1)
    S = np.zeros_like(D)
    S[:H - k + 1, :W - k + 1] = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return S


def _img(crop=64, st=4, sigma=25.0):
    lf = Hh.source_lf(crop=crop)
    _, noisy = Hh.noisy_lf(lf, sigma)
    return noisy.reshape(9, 3, crop, crop)


@pytest.mark.parametrize("k,nDisp", [(8, 2), (16, 3)])
def test_disparity_search_against_brute_force_ssd(k, nDisp):
    """precompute_BM_stereo (core:3479-3611): arg-min over (2 nDisp+1)^2 displacements of the true SSD between the patch
    at a position of image 1 and the displaced patch of image 2; ties in scan order (dj outer, di inner)."""
    lf = _img(72)
    img1, img2 = np.ascontiguousarray(lf[4, 0]), np.ascontiguousarray(lf[5, 0])
    H, W = img1.shape
    tau = 3000.0
    best, shape = np.zeros(W * H, np.uint32), np.zeros(W * H, np.uint8)
    assert O.lib().orc_bm_stereo(img1.reshape(-1), img2.reshape(-1), W, H, k, nDisp, tau, best, shape) == 0
    best, shape = best.reshape(H, W), shape.reshape(H, W)
    rows, cols = slice(nDisp, H - nDisp - k + 1), slice(nDisp, W - nDisp - k + 1)
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    cand = []   # (order, di, dj, S)
    for dj in range(-nDisp, nDisp + 1):
        for di in range(-nDisp, nDisp + 1):
            D = np.zeros((H, W))
            ys, xs = slice(nDisp, H - nDisp), slice(nDisp, W - nDisp)            # the band the reference fills
            D[ys, xs] = (b[nDisp + di:H - nDisp + di, nDisp + dj:W - nDisp + dj] - a[ys, xs]) ** 2
            cand.append((di, dj, _box(D, k)[rows, cols]))
    S = np.stack([c[2] for c in cand])                                        # scan order along axis 0
    o = np.argsort(S, axis=0, kind="stable")
    s0, s1 = np.take_along_axis(S, o[:1], 0)[0], np.take_along_axis(S, o[1:2], 0)[0]
    pos = np.arange(H * W).reshape(H, W)[rows, cols]
    di0 = np.array([c[0] for c in cand])[o[0]]
    dj0 = np.array([c[1] for c in cand])[o[0]]
    expect = pos + di0 * W + dj0
    clear = (s1 - s0) > 1e-4 * np.maximum(s1, 1.0)        # float32 integral images may re-order closer calls
    assert clear.mean() > 0.98
    assert np.array_equal(best[rows, cols][clear], expect[clear])
    thr = tau * k * k
    sure = np.abs(s0 - thr) > 1e-4 * thr
    assert np.array_equal(shape[rows, cols][sure] != 0, (s0 < thr)[sure])


@pytest.mark.parametrize("k,N,nSim,nDisp,p,tau", [(8, 8, 6, 2, 4, 3000.0), (16, 4, 5, 3, 3, 3000.0), (8, 16, 7, 1, 5, 800.0)])
def test_self_similarity_search_against_brute_force_ssd(k, N, nSim, nDisp, p, tau):
    """precompute_BM (core:3301-3461) from its definition.  For a displacement d = (di >= 0, dj) the table is
    S_d(i, j) = sum over the k x k box at (i, j) of D_d, with D_d(y, x) = (img[y+di][x+dj] - img[y][x])^2 inside the band
    [nHW, dim-nHW) and 0 outside (quirk 6: patches reaching the far band are under-estimated); S_d exists for (i, j) in
    the band, everything else reads 2*threshold.  A forward candidate r + d is tested and scored with S_d(r); a backward
    candidate r - d is TESTED with S_d(r) and SCORED with S_d(r - d) (2*threshold if r - d is outside the band).  Scan
    order: dj outer, di = 0..nSim then di = -nSim..-1.  nSx = N if enough candidates pass, else the largest power of
    two; the nSx best by (score, scan order); a single survivor is stored twice."""
    img = np.ascontiguousarray(_img(80)[4, 0])
    H, W = img.shape
    nHW = nSim + nDisp
    lib = O.lib()
    buf = np.zeros(H, np.uint32)
    nr = lib.orc_ind_initialize(H - k + 1, nHW, p, buf.ctypes.data); rws = buf[:nr].copy()
    nc = lib.orc_ind_initialize(W - k + 1, nHW, p, buf.ctypes.data); cls = buf[:nc].copy()
    refs = np.array([r * W + c for r in rws for c in cls], np.uint32)
    idx = np.zeros((len(refs), N), np.uint32)
    cnt = np.zeros(len(refs), np.uint32)
    assert lib.orc_bm_self(img.reshape(-1), W, H, k, N, nHW, nSim, tau, refs, len(refs), idx.reshape(-1), cnt) == 0
    thr = tau * k * k
    a = img.astype(np.float64)
    band = np.zeros((H, W), bool)
    band[nHW:H - nHW, nHW:W - nHW] = True
    tables = {}
    for di in range(0, nSim + 1):
        for dj in range(-nSim, nSim + 1):
            D = np.zeros((H, W))
            ys, xs = slice(nHW, H - nHW), slice(nHW, W - nHW)
            D[ys, xs] = (a[nHW + di:H - nHW + di, nHW + dj:W - nHW + dj] - a[ys, xs]) ** 2
            T = np.full((H, W), 2 * thr)
            Sb = np.zeros((H, W))
            c = np.zeros((H + k + 1, W + k + 1))
            c[1:H + 1, 1:W + 1] = D.cumsum(0).cumsum(1)
            c[H + 1:, :] = c[H:H + 1, :]                 # zero band continues past the array
            c[:, W + 1:] = c[:, W:W + 1]
            ii, jj = np.arange(H)[:, None], np.arange(W)[None, :]
            Sb = c[ii + k, jj + k] - c[ii, jj + k] - c[ii + k, jj] + c[ii, jj]
            T[band] = Sb[band]
            tables[(di, dj)] = T
    checked = 0
    for r, k_r in enumerate(refs):
        ri, rj = int(k_r) // W, int(k_r) % W
        cands = []
        for dj in range(-nSim, nSim + 1):
            for di in range(0, nSim + 1):
                v = tables[(di, dj)][ri, rj]
                cands.append((v, v, (ri + di) * W + rj + dj))
            for di in range(-nSim, 0):
                T = tables[(-di, -dj)]
                ci, cj = ri + di, rj + dj
                cands.append((T[ri, rj], T[ci, cj], ci * W + cj))
        tests = np.array([c[0] for c in cands]); scores = np.array([c[1] for c in cands]); posn = np.array([c[2] for c in cands])
        if (np.abs(tests - thr) < 1e-4 * thr).any():
            continue                                       # a threshold decision within float rounding
        ok = tests < thr
        n_ok = int(ok.sum())
        nSx = N if n_ok >= N else (1 << (n_ok.bit_length() - 1) if n_ok else 1)
        if n_ok == 0:
            assert cnt[r] == 2 and idx[r, 0] == idx[r, 1] == k_r
            checked += 1
            continue
        sc, ps = scores[ok], posn[ok]
        o = np.argsort(sc, kind="stable")
        if nSx < n_ok and sc[o[nSx]] - sc[o[nSx - 1]] <= 1e-4 * max(sc[o[nSx]], 1.0):
            continue                                       # the cut falls between two near-tied candidates
        expect = set(ps[o[:nSx]].tolist())
        got = idx[r, :cnt[r]].tolist()
        assert cnt[r] == (2 if nSx == 1 else nSx)
        assert set(got) == expect, (r, sorted(got), sorted(expect))
        assert got[0] == ps[o[0]] or sc[o[1]] - sc[o[0]] <= 1e-4 * max(sc[o[1]], 1.0)    # best first (the reference patch itself)
        checked += 1
    assert checked > 0.9 * len(refs)


# ------------------------------------------------------------------------------------------------------------------
# shape-adaptive DCT
# ------------------------------------------------------------------------------------------------------------------
def _dct_1d(x):
    """FFTW REDFT10 times the reference's coef_norm (core:3229-3252) == orthonormal DCT-II times sqrt(2)."""
    n = len(x)
    y = scipy.fft.dct(np.asarray(x, np.float64), type=2)          # 2 sum x_j cos(pi (j + 1/2) k / n)
    cn = np.full(n, np.sqrt(2.0) / np.sqrt(n)); cn[0] = 1.0 / np.sqrt(n)
    return y * cn


def _sadct_model(v, mask, aw, ah):
    g = np.array(v, np.float64).reshape(ah, aw)
    m = np.array(mask).reshape(ah, aw) != 0
    mcol = np.zeros((ah, aw), bool)
    for s in range(ah):                       # rows: in-shape entries, transformed, compacted to the row start
        sel = g[s, m[s]]
        n = len(sel)
        if n == 1:
            g[s, 0] = sel[0]
        elif n > 1:
            g[s, :n] = _dct_1d(sel)
        mcol[s, :n] = True
    mdct = np.zeros((ah, aw), bool)
    for t in range(aw):                       # columns of the compacted rows
        sel = g[mcol[:, t], t]
        n = len(sel)
        if n == 1:
            g[0, t] = sel[0]
        elif n > 1:
            g[:n, t] = _dct_1d(sel)
        mdct[:n, t] = True
    return np.where(mdct, g * (0.5 / np.sqrt(2.0)), 0.0).reshape(-1), mdct.reshape(-1)


@pytest.mark.parametrize("aw,ah", [(3, 3), (5, 5), (3, 5)])
def test_sadct_forward_values_on_every_mask(aw, ah):
    rng = np.random.default_rng(aw * 10 + ah)
    A = aw * ah
    masks = range(1, 1 << A) if A <= 9 else [int(x) for x in rng.integers(1, 1 << A, size=600)]
    lib = O.lib()
    for bits in masks:
        mask = np.array([(bits >> i) & 1 for i in range(A)], np.uint32)
        v = rng.uniform(-200, 200, A).astype(np.float32)
        out = v.copy()
        mdct = np.zeros(A, np.uint32)
        lib.orc_sadct_forward(out, mask, aw, ah, mdct)
        exp, emd = _sadct_model(v, mask, aw, ah)
        assert np.array_equal(mdct != 0, emd), bits
        np.testing.assert_allclose(out[emd], exp[emd], rtol=2e-5, atol=2e-4)
        assert not out[~emd].any()


# ------------------------------------------------------------------------------------------------------------------
# 5th-dimension filters
# ------------------------------------------------------------------------------------------------------------------
def _haar_matrix(n):
    """Orthonormal multi-level Haar of lib_transforms.cpp:403-471: averages first, then details, level by level."""
    M = np.eye(n)
    m = n
    while m > 1:
        L = np.eye(n)
        L[:m, :m] = 0
        for i in range(m // 2):
            L[i, 2 * i] = L[i, 2 * i + 1] = 1 / np.sqrt(2)
            L[m // 2 + i, 2 * i] = 1 / np.sqrt(2); L[m // 2 + i, 2 * i + 1] = -1 / np.sqrt(2)
        M = L @ M
        m //= 2
    return M


@pytest.mark.parametrize("nSx", [2, 4, 8, 16])
@pytest.mark.parametrize("masked", [False, True])
def test_hard_threshold_haar_slab_against_numpy_model(nSx, masked):
    rng = np.random.default_rng(nSx + 100 * masked)
    A, Cc = 9, 3
    X = rng.normal(0, 60, (Cc, A, nSx)).astype(np.float32)
    sigma = np.array([14.4, 17.7, 15.3], np.float32)
    lam = 2.7
    md = (rng.integers(0, 2, A).astype(np.uint32) | np.eye(1, A, 4, dtype=np.uint32)[0]) if masked else None
    got = X.copy()
    w = np.zeros(Cc, np.float32)
    O.lib().orc_ht_filter_slab(got.reshape(-1), nSx, A, Cc, sigma, lam, w, md.ctypes.data if masked else None, O.HAAR)
    M = _haar_matrix(nSx)
    Y = X.astype(np.float64) @ M.T                               # forward along n
    T = (lam * sigma.astype(np.float64) * np.sqrt(2.0))[:, None, None]
    keep = np.abs(Y) > T
    near = np.abs(np.abs(Y) - T) < 1e-4 * T                       # decisions within float rounding of the threshold
    sel = np.ones(A, bool) if not masked else md != 0
    Yf = np.where(keep, Y, 0.0)
    Yf[:, ~sel] = Y[:, ~sel]                                      # out-of-shape SAIs are left untouched (quirk 10)
    exp = Yf @ M                                                  # orthonormal: inverse = transpose
    if not near[:, sel].any():
        np.testing.assert_allclose(got, exp, rtol=1e-5, atol=2e-4)
        assert np.array_equal(w, keep[:, sel].sum(axis=(1, 2)).astype(np.float32))


@pytest.mark.parametrize("nSx", [1, 2, 8, 16])
@pytest.mark.parametrize("masked", [False, True])
def test_wiener_haar_slab_against_numpy_model(nSx, masked):
    rng = np.random.default_rng(nSx + 7 + 100 * masked)
    A, Cc = 9, 3
    Xo = rng.normal(0, 60, (Cc, A, nSx)).astype(np.float32)
    Xe = (Xo + rng.normal(0, 10, Xo.shape)).astype(np.float32)
    sigma = np.array([14.4, 17.7, 15.3], np.float32)
    md = (rng.integers(0, 2, A).astype(np.uint32) | np.eye(1, A, 4, dtype=np.uint32)[0]) if masked else None
    go, ge = Xo.copy(), Xe.copy()
    w = np.zeros(Cc, np.float32)
    O.lib().orc_wiener_filter_slab(go.reshape(-1), ge.reshape(-1), nSx, A, Cc, sigma, w, md.ctypes.data if masked else None, O.HAAR)
    M = _haar_matrix(nSx) if nSx > 1 else np.eye(1)
    Yo, Ye = Xo.astype(np.float64) @ M.T, Xe.astype(np.float64) @ M.T
    s2 = (sigma.astype(np.float64) ** 2)[:, None, None]
    v = Ye ** 2 / (Ye ** 2 + s2)
    sel = np.ones(A, bool) if not masked else md != 0
    F = np.where(sel[None, :, None], Yo * v, Ye)                  # out-of-shape: the pilot's coefficients stay
    exp = F @ M
    np.testing.assert_allclose(ge, exp, rtol=2e-5, atol=5e-4)
    np.testing.assert_allclose(w, v[:, sel].sum(axis=(1, 2)), rtol=1e-5)
