#!/usr/bin/env python3
"""Hard-threshold decisions against the oracle, group by group: for the hard-thresholding parity cases (tests/test_gpu_parity.py
PASS_CASES) the number of (group, channel) pairs whose survivor count differs from the oracle's, and by how much.  A/B tool for
changes of the transform arithmetic (LFBM5D_HIP_LIB selects the library).
usage: python tools/flip_count.py [case-name-substring ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
import lfbm5d_amd as L  # noqa: E402


def full(ctx, seeds):
    """the headline window (3x3x512x512 -> 560x560, README HT parameters: 864 M coefficients per pass) on the benchmark's synthetic
    light field, one pass per noise seed"""
    from lfbm5d_amd import synth
    pk, sigma, H, W = Hh.README_HT, 25.0, 512, 512
    lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
    for seed in seeds:
        noisy = lf + sigma * np.random.default_rng(seed).standard_normal(lf.shape).astype(np.float32)
        win, Wb, Hb = Hh.padded_window(np.ascontiguousarray(noisy.reshape(9, -1)), W, H, 3, pk[1] + pk[2])
        num_o, den_o, st = Hh.oracle_pass(1, sigma, pk, win, None, Wb, Hb, 3)
        num_g, den_g = T.gpu_pass(ctx, 1, sigma, pk, win, None, Wb, Hb, 3)
        R = len(ctx.last_bm(pk[0], 9, Wb * Hb)[0])
        w_o, w_g = O.last_weights(R, 3), ctx.last_weights(R, 3)
        sig = np.zeros(4, np.float32)
        O.lib().orc_sigma_table(sigma, 3, O.OPP, sig)
        cnt_o = np.where(w_o == 1.0, 0.0, 1.0 / (w_o.astype(np.float64) * sig[:3].astype(np.float64) ** 2))
        cnt_g = np.where(w_g == 1.0, 0.0, 1.0 / (w_g.astype(np.float64) * sig[:3].astype(np.float64) ** 2))
        d = np.rint(cnt_g - cnt_o)
        print(f"full 560x560 seed {seed}: groups x channels {R * 3}  coefficients {int(st.stack_patches) * 9 * 256 * 3}  flipped {int((d != 0).sum())} "
              f"(max |delta| {int(np.abs(d).max())})  survivors {int(np.rint(cnt_o).sum())}", flush=True)


def main():
    ctx = L.Context(0)
    pats = sys.argv[1:]
    if pats and pats[0] == "full":
        full(ctx, [int(x) for x in pats[1:]] or [1])
        ctx.close()
        return
    for case in T.PASS_CASES:
        name, step, sigma, pk, crop, useSD = case
        if step != 1 or useSD or pk[7] != "haar" or (pats and not any(p in name for p in pats)):
            continue
        win, Wb, Hb, Cc = T.window(sigma, pk, crop)
        num_o, den_o, st = Hh.oracle_pass(step, sigma, pk, win, None, Wb, Hb, Cc)
        num_g, den_g = T.gpu_pass(ctx, step, sigma, pk, win, None, Wb, Hb, Cc)
        refs = ctx.last_bm(pk[0], 9, Wb * Hb)[0]
        R = len(refs)
        w_o, w_g = O.last_weights(R, Cc), ctx.last_weights(R, Cc)
        sig = np.zeros(4, np.float32)
        O.lib().orc_sigma_table(sigma, Cc, O.OPP, sig)
        cnt_o = np.where(w_o == 1.0, 0.0, 1.0 / (w_o.astype(np.float64) * sig[:Cc].astype(np.float64) ** 2))
        cnt_g = np.where(w_g == 1.0, 0.0, 1.0 / (w_g.astype(np.float64) * sig[:Cc].astype(np.float64) ** 2))
        d = np.rint(cnt_g - cnt_o)
        eo, eg = Hh.estimate(num_o, den_o, win), Hh.estimate(num_g, den_g, win)
        print(f"{name:24s} groups x channels {R * Cc:6d}  coefficients {int(st.stack_patches) * 9 * pk[3] ** 2 * Cc:10d}  flipped {int((d != 0).sum()):4d} "
              f"(max |delta| {int(np.abs(d).max())})  survivors {int(np.rint(cnt_o).sum()):9d}  max |est diff| {np.abs(eo - eg).max():.2e}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
