#!/usr/bin/env python3
"""Per-kernel-class time of one window pass on a WIDE angular window (aswSize > 1): aw x aw SAIs of 256 x 256 (304^2 padded), README
parameters, both steps; round-5 review item 7.  usage: python tools/wide_window_time.py <aw> [steps: 1 / 2 / 12] [reps]
(option band_mb through LFBM5D_BAND_MB caps the filt buffer: the pass then runs band by band)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    aw = int(sys.argv[1])
    steps = sys.argv[2] if len(sys.argv) > 2 else "12"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    H = W = 256
    sigma = 25.0
    A = aw * aw
    ctx = L.Context(0)
    lf = synth.make_lf(aw, aw, H, W).reshape(A, 3, H, W).astype(np.float32)
    lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    for step, pk in ((1, (8, 18, 6, 16, 4, "id", "sadct", "haar")), (2, (16, 18, 6, 8, 4, "dct", "sadct", "haar"))):
        if str(step) not in steps:
            continue
        P = core.make_params(sigma, 2.7, *pk)
        nHW = pk[1] + pk[2]
        pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
        Hb, Wb = pad.shape[2:]
        noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(A, -1)).cuda()
        basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
        num, den = torch.zeros_like(noisy), torch.zeros_like(noisy)
        mask, proc = np.ones(A, np.uint32), np.zeros(A, np.uint32)
        for it in range(reps + 1):
            if it == 1:
                torch.cuda.synchronize()
                ctx.reset_stats()
            ctx.core_pass(step, P, aw, aw, Wb, Hb, 3, noisy, basic, num, den, mask, proc, A // 2, A // 2)
        torch.cuda.synchronize()
        s = ctx.stats()
        filt_gb = s.stack_patches / reps * A * pk[3] * pk[3] * 3 * 4 / 2 ** 30
        print(f"{aw}x{aw} window, step {step} (k {pk[3]}, N {pk[0]}), band_mb {os.environ.get('LFBM5D_BAND_MB', '-')}: bm {s.ms_bm / reps:.2f} group {s.ms_group / reps:.2f} "
              f"agg {s.ms_aggregate / reps:.2f} ms per pass; filt {filt_gb:.1f} GiB; checksum {float(num.double().sum()):.6e} {float(den.double().sum()):.6e}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
