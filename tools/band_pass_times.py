#!/usr/bin/env python3
"""What a spatial band costs in time (round 6; input of tools/scale_model.py's band model): the two-step job on the first R rows of
every SAI of a light field, R = the band heights of 1 / 2 / 4 bands with the default halo, against the whole field -- one GPU, the
library's default lanes, min of 3 runs -- next to the model's factor (R + 2 nHW) / (H + 2 nHW).  Then the accuracy of the banded
job at the headline size: emulate_world = 8 with spatial_bands = 2 against the one-GPU result.
usage: python tools/band_pass_times.py [ah aw H W] > profiles/<tag>_band_pass_times.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    ah, aw, H, W = a[:4] if len(a) >= 4 else (17, 17, 512, 512)
    sigma, halo, nhw = 25.0, 40, 24
    A = ah * aw
    clean = synth.make_lf(ah, aw, H, W).reshape(A, 3, H, W).astype(np.float32)
    noisy = synth.add_noise_mt19937(clean.reshape(A, -1), sigma, seed=1).reshape(A, 3, H, W)
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    mask = np.ones(A, np.uint32)
    ctx = L.Context(0)

    def job(rows, reps=3):
        src = torch.from_numpy(np.ascontiguousarray(noisy[:, :, :rows]).reshape(A, -1)).cuda()
        d_n, d_b, d_d = torch.empty_like(src), torch.zeros_like(src), torch.zeros_like(src)
        best = 1e9
        for _ in range(reps + 1):
            d_n.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.denoise(P1, P2, d_n, mask, d_b, d_d, L.ROWMAJOR, aw, ah, 1, 1, W, rows, 3)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3, d_b, d_d

    print(f"# {ah}x{aw}x{H}x{W}, sigma 25, README parameters, one GPU, lanes {ctx.get_option('lanes')}; halo {halo}, nHW {nhw}")
    print("rows   job_ms   measured/whole   model (rows + 2 nHW)/(H + 2 nHW)")
    t_full, b_full, d_full = job(H)
    for S in (1, 2, 4):
        rows = H if S == 1 else min(H, H // S + (2 if S > 2 else 1) * halo)
        t, _, _ = job(rows) if S > 1 else (t_full, None, None)
        print(f"{rows:4d} {t:8.1f} {t / t_full:12.3f} {(rows + 2 * nhw) / (H + 2 * nhw):12.3f}     # widest band of {S}")

    def psnr(x):
        mse = ((x.double() - torch.from_numpy(clean.reshape(A, -1)).cuda().double()) ** 2).mean(dim=1)
        return float((20 * torch.log10(255.0 / torch.sqrt(mse))).mean().item())
    print(f"# banded job against one GPU: one GPU basic {psnr(b_full):.4f} dB, denoised {psnr(d_full):.4f} dB")
    for (world, S) in ((8, 2), (8, 4), (4, 2)):
        ctx.set_option("emulate_world", world)
        ctx.set_option("spatial_bands", S)
        src = torch.from_numpy(noisy.reshape(A, -1)).cuda()
        d_b, d_d = torch.zeros_like(src), torch.zeros_like(src)
        ctx.denoise(P1, P2, src, mask, d_b, d_d, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
        torch.cuda.synchronize()
        dd = (d_d - d_full).abs()
        print(f"emulate_world {world}, spatial_bands {S}: basic {psnr(d_b):.4f} dB ({psnr(d_b) - psnr(b_full):+.1e}), denoised {psnr(d_d):.4f} dB "
              f"({psnr(d_d) - psnr(d_full):+.1e}); mean |d| {float(dd.mean()):.2e}, pixels off by > 1 grey level {int((dd > 1).sum())} of {dd.numel()}")
    ctx.close()


if __name__ == "__main__":
    main()
