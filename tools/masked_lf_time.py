#!/usr/bin/env python3
"""Time of the two-step job on a light field with empty SAIs (a lenslet light field whose corner views are missing): 9x9x512x512,
the 2x2 SAIs of every corner empty.  Windows with an empty SAI run the shape-adaptive angular transform in every group
(bm5d.cpp:276-280) -- since round 4 in kernels of their own.
usage: python tools/masked_lf_time.py [reps] [lenslet]     lenslet: 15x15x434x625, sigma 10, the README's lenslet parameters
(bior / sadct / haar, dct / sadct / haar), the three SAIs nearest every corner empty"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    lenslet = len(sys.argv) > 2 and sys.argv[2] == "lenslet"
    ah = aw = 15 if lenslet else 9
    H, W = (434, 625) if lenslet else (512, 512)
    sigma = 10.0 if lenslet else 25.0
    lf = synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1).astype(np.float32)
    noisy_h = lf + sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "bior" if lenslet else "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    ctx = L.Context(0)
    for corners in (0, 2):
        mask = np.ones((ah, aw), np.uint32)
        for s in range(corners):
            for t in range(corners):
                if lenslet and s + t > 1:
                    continue   # (0,0), (0,1), (1,0) of every corner
                mask[s, t] = mask[s, aw - 1 - t] = mask[ah - 1 - s, t] = mask[ah - 1 - s, aw - 1 - t] = 0
        mask = mask.reshape(-1)
        src = noisy_h.copy()
        src[mask == 0] = 0
        d_n0 = torch.from_numpy(src).cuda()
        d_n, d_b, d_o = torch.empty_like(d_n0), torch.zeros_like(d_n0), torch.zeros_like(d_n0)
        for it in range(reps + 1):
            d_n.copy_(d_n0)
            torch.cuda.synchronize()
            ctx.reset_stats()
            t0 = time.perf_counter()
            ctx.denoise(P1, P2, d_n, mask, d_b, d_o, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
        s = ctx.stats()
        print(f"{ah}x{aw}x{H}x{W}, {int((mask == 0).sum())} empty SAIs: {t * 1e3:7.1f} ms  windows {s.windows}  groups {s.groups} of them shape-adaptive {s.sadct_groups}  "
              f"checksum {float(d_o.double().sum()):.6e}", flush=True)


if __name__ == "__main__":
    main()
