#!/usr/bin/env python3
"""Per-kernel-class time of one centre-window core pass at the headline window size (3x3 SAIs of
512x512, padded to 560x560), both steps, README parameters.  No oracle, GPU only: a quick
A/B tool for kernel work.
usage: python tools/pass_time.py [reps] [H] [tau_2D of the HT step: id / bior / dct] [tau_2D of the Wiener step]
       [N of the HT step] [N of the Wiener step]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    W = H
    sigma = float(os.environ.get("PASS_TIME_SIGMA", "25"))
    lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
    rng = np.random.default_rng(1)
    lf += sigma * rng.standard_normal(lf.shape).astype(np.float32)
    ctx = L.Context(0)
    ht2d = sys.argv[3] if len(sys.argv) > 3 else "id"
    wi2d = sys.argv[4] if len(sys.argv) > 4 else "dct"
    n1 = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    n2 = int(sys.argv[6]) if len(sys.argv) > 6 else 16
    for step, pk in ((1, (n1, 18, 6, 16, 4, ht2d, "sadct", "haar")), (2, (n2, 18, 6, 8, 4, wi2d, "sadct", "haar"))):
        P = core.make_params(sigma, 2.7, *pk)
        nHW = pk[1] + pk[2]
        pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
        Hb, Wb = pad.shape[2:]
        noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
        basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
        num = torch.zeros_like(noisy)
        den = torch.zeros_like(noisy)
        mask = np.ones(9, np.uint32)
        proc = np.zeros(9, np.uint32)
        for e in [int(x) for x in os.environ.get("PASS_TIME_EMPTY", "").split(",") if x]:   # empty SAIs: every group shape-adaptive (bm5d.cpp:276-280)
            mask[e] = 0; proc[e] = 1; noisy[e] = 0
        for it in range(reps + 1):
            if it == 1:
                torch.cuda.synchronize()
                ctx.reset_stats()
            ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic, num, den, mask, proc, 4, 4)
        torch.cuda.synchronize()
        s = ctx.stats()
        print(f"step {step} {Wb}x{Hb}: bm {s.ms_bm / reps:.3f} group {s.ms_group / reps:.3f} agg {s.ms_aggregate / reps:.3f} "
              f"other {s.ms_other / reps:.3f} ms/pass; groups {s.groups // reps} (shape-adaptive {s.sadct_groups // reps}, mean stack {s.stack_patches / max(1, s.groups):.2f}) "
              f"checksum {float(num.double().sum()):.6e}")


if __name__ == "__main__":
    main()
