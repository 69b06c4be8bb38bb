#!/usr/bin/env python3
"""Experiment: the group + aggregation pair of one 560x560 window pass run in n bands of reference-patch rows one after the other
(the library's row shards), with each band's filtered patches (a) at their place in the 3.46 / 1.78 GB buffer, (b) at the start
of the buffer (LFBM5D_FILT_BAND=1: the bands reuse the same <= 3.46 / n GB).  Times: sum of the bands' kernels (HIP events).
Results are NOT a denoised window (shards > 0 start from zeroed sums); only the times matter."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402

H = W = 512
sigma = 25.0
lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
ctx = L.Context(0)
for step, pk in ((1, (8, 18, 6, 16, 4, "id", "sadct", "haar")), (2, (16, 18, 6, 8, 4, "dct", "sadct", "haar"))):
    P = core.make_params(sigma, 2.7, *pk)
    nHW = pk[1] + pk[2]
    pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
    Hb, Wb = pad.shape[2:]
    noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
    basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
    num, den = torch.zeros_like(noisy), torch.zeros_like(noisy)
    mask, proc = np.ones(9, np.uint32), np.zeros(9, np.uint32)
    for nb in [int(x) for x in sys.argv[1:]] or [1, 4, 8, 16, 24, 32]:
        reps = 3
        for it in range(reps + 1):
            if it == 1:
                torch.cuda.synchronize()
                ctx.reset_stats()
            for r in range(nb):
                ctx.set_shard(r, nb)
                ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic, num, den, mask, proc, 4, 4)
        torch.cuda.synchronize()
        s = ctx.stats()
        print(f"step {step} bands {nb:3d}: group {s.ms_group / reps:.3f} agg {s.ms_aggregate / reps:.3f} ms (sum over the bands)  band of filt {(3.46 if step == 1 else 1.78) / nb * 1e3:.0f} MB", flush=True)
    ctx.set_shard(0, 1)
