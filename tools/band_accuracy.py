#!/usr/bin/env python3
"""What spatial bands cost in accuracy (round 6, multi-GPU design study): the two-step job on a horizontal BAND of every SAI
(its rows plus a halo) against the same rows of the job on the whole light field.  One GPU, no ranks involved.
usage: python tools/band_accuracy.py [ah aw H W] [bands] [halo ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def run(ctx, P1, P2, noisy, ah, aw, H, W):
    d_n = torch.from_numpy(np.ascontiguousarray(noisy)).cuda()
    d_b, d_d = torch.zeros_like(d_n), torch.zeros_like(d_n)
    mask = np.ones(ah * aw, np.uint32)
    ctx.denoise(P1, P2, d_n, mask, d_b, d_d, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
    return d_b.cpu().numpy(), d_d.cpu().numpy()


def main():
    a = [int(x) for x in sys.argv[1:]]
    ah, aw, H, W = a[:4] if len(a) >= 4 else (9, 9, 512, 512)
    S = a[4] if len(a) > 4 else 2
    halos = a[5:] if len(a) > 5 else [40, 64, 96]
    sigma = 25.0
    clean = synth.make_lf(ah, aw, H, W).reshape(ah * aw, 3, H, W).astype(np.float32)
    noisy = synth.add_noise_mt19937(clean.reshape(ah * aw, -1), sigma, seed=1).reshape(ah * aw, 3, H, W)
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    ctx = L.Context(0)
    b_full, d_full = run(ctx, P1, P2, noisy.reshape(ah * aw, -1), ah, aw, H, W)
    b_full, d_full = b_full.reshape(ah * aw, 3, H, W), d_full.reshape(ah * aw, 3, H, W)

    def psnr(x, ref):
        mse = ((x.astype(np.float64) - ref) ** 2).reshape(x.shape[0], -1).mean(axis=1)
        return float((20 * np.log10(255.0 / np.sqrt(mse))).mean())
    print(f"# {ah}x{aw}x{H}x{W}, sigma 25, README parameters, {S} bands; whole light field: basic {psnr(b_full, clean):.4f} dB, denoised {psnr(d_full, clean):.4f} dB")
    print("halo  band  rows(crop)  psnr_full(band rows)  psnr_banded   delta_dB   mean|d|   px>1.0   px>0.1   pixels")
    for halo in halos:
        tot = np.zeros(3)
        stitched = np.zeros_like(d_full)
        for b in range(S):
            y0, y1 = b * H // S, (b + 1) * H // S
            c0, c1 = max(0, y0 - halo), min(H, y1 + halo)
            crop = np.ascontiguousarray(noisy[:, :, c0:c1, :])
            _, d_c = run(ctx, P1, P2, crop.reshape(ah * aw, -1), ah, aw, c1 - c0, W)
            d_c = d_c.reshape(ah * aw, 3, c1 - c0, W)[:, :, y0 - c0:y1 - c0, :]
            stitched[:, :, y0:y1, :] = d_c
            ref, cl = d_full[:, :, y0:y1, :], clean[:, :, y0:y1, :]
            d = np.abs(d_c.astype(np.float64) - ref)
            print(f"{halo:4d} {b:5d} {c1 - c0:10d} {psnr(ref, cl):20.4f} {psnr(d_c, cl):12.4f} {psnr(d_c, cl) - psnr(ref, cl):+10.1e} {d.mean():9.2e} {int((d > 1).sum()):8d} {int((d > 0.1).sum()):8d} {d.size:8d}")
        print(f"{halo:4d}   all {'':10s} {psnr(d_full, clean):20.4f} {psnr(stitched, clean):12.4f} {psnr(stitched, clean) - psnr(d_full, clean):+10.1e}")
    ctx.close()


if __name__ == "__main__":
    main()
