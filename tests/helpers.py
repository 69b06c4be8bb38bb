"""Shared helpers of the parity tests: build padded windows like run_bm5d_* does and drive the
oracle (checker) and the HIP path (product) on the same inputs."""
import ctypes as C
import os

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# parameter tuples: (N, nSim, nDisp, k, p, tau_2D, tau_4D, tau_5D)
README_HT = (8, 18, 6, 16, 4, "id", "sadct", "haar")       # README.md:50
README_WIEN = (16, 18, 6, 8, 4, "dct", "sadct", "haar")
C4_HT = (8, 18, 6, 16, 4, "bior", "sadct", "haar")          # BASELINE.json configs[3]
C5_HT = (1, 18, 3, 16, 3, "bior", "sadct", "haar")          # BASELINE.json configs[4]
C5_WIEN = (8, 18, 3, 8, 3, "dct", "sadct", "haar")


def source_lf(crop=None):
    lf = np.load(os.path.join(GOLDEN, "sourceLF_3x3_256_u8.npy"))
    if crop:
        ch, cw = (crop, crop) if np.isscalar(crop) else crop     # (rows, columns) for non-square SAIs
        lf = lf[:, :, :ch, :cw]
    return np.ascontiguousarray(lf)


def noisy_lf(lf_u8, sigma, seed=1):
    A = lf_u8.shape[0]
    clean = np.ascontiguousarray(lf_u8.astype(np.float32)).reshape(A, -1)
    return clean, O.add_noise_lf(clean, sigma, seed=seed)


def padded_window(arr, W, H, Cc, nHW, cs=O.OPP, color=True):
    """[A][C*H*W] RGB -> colour-transformed, mirror-padded [A][C*Hb*Wb] (bm5d.cpp:133, :261)."""
    lib = O.lib()
    A = arr.shape[0]
    Wb, Hb = W + 2 * nHW, H + 2 * nHW
    out = np.zeros((A, Cc * Wb * Hb), np.float32)
    for st in range(A):
        im = np.ascontiguousarray(arr[st]).copy()
        if color:
            lib.orc_color_transform(im, cs, W, H, Cc, 1)
        lib.orc_symetrize(im, out[st], W, H, Cc, nHW)
    return out, Wb, Hb


def oracle_pass(step, sigma, pk, win, basic, Wb, Hb, Cc, num=None, den=None, mask=None, proc=None,
                rows=(0, -1), cst=4, pst=4, useSD=0, aw=3):
    A = win.shape[0]
    num = np.zeros_like(win) if num is None else num
    den = np.zeros_like(win) if den is None else den
    mask = np.ones(A, np.uint32) if mask is None else mask
    proc = np.zeros(A, np.uint32) if proc is None else proc
    st = O.Stats()
    P = O.make_params(sigma, 2.7, *pk, useSD=useSD)
    rc = O.lib().orc_pass(step, C.byref(P), aw, aw, Wb, Hb, Cc, win.reshape(-1),
                          basic.ctypes.data if basic is not None else None, num.reshape(-1), den.reshape(-1),
                          mask, proc, cst, pst, rows[0], rows[1], C.byref(st))
    assert rc == 0
    return num, den, st


def estimate(num, den, sub):
    return np.where(den != 0, num / np.where(den != 0, den, 1), sub)


def tau_match(sigma, Cc, step):
    sig = np.zeros(3, np.float32)
    O.lib().orc_sigma_table(sigma, Cc, O.OPP, sig)
    return (3.0 if Cc == 1 else 1.0) * ((3000 if step == 1 else 2000) if sig[0] < 35 else 5000)


def textured_lf(ah, aw, H, W, disparity=1):
    """A light field with natural texture for multi-window tests: SAI (s, t) is the centre SAI of
    tests/golden/sourceLF (a photograph) cropped at an offset that moves `disparity` pixels per view, so every SAI
    pair has a real, sub-nDisp disparity and block matching has unambiguous minima (flat synthetic shapes do not)."""
    src = np.load(os.path.join(GOLDEN, "sourceLF_3x3_256_u8.npy"))[4]           # [3][256][256]
    y0 = (256 - H - disparity * (ah - 1)) // 2
    x0 = (256 - W - disparity * (aw - 1)) // 2
    assert y0 >= 0 and x0 >= 0
    out = np.zeros((ah * aw, 3, H, W), np.uint8)
    for s in range(ah):
        for t in range(aw):
            out[s * aw + t] = src[:, y0 + disparity * s:y0 + disparity * s + H, x0 + disparity * t:x0 + disparity * t + W]
    return out
