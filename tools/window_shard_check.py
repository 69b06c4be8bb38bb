#!/usr/bin/env python3
"""Window-sharded steps, all ranks played on one GPU (LFBM5D_EMULATE_WORLD): PSNR against the
single-GPU (reference-order) result, and the planned window sequence against the data-driven one.
usage: python tools/window_shard_check.py [ah aw H W]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def run(ctx, noisy0, ah, aw, H, W, sigma):
    asize = ah * aw
    mask = np.ones(asize, np.uint32)
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    noisy = noisy0.clone()
    basic = torch.zeros_like(noisy)
    den = torch.zeros_like(noisy)
    ctx.step1(P1, noisy, mask, basic, L.ROWMAJOR, aw, ah, 1, W, H, 3)
    w1 = ctx.last_windows()
    ctx.step2(P2, noisy, mask, basic, den, L.ROWMAJOR, aw, ah, 1, W, H, 3)
    return basic, den, w1


def main():
    ah, aw, H, W = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (5, 5, 128, 128)
    sigma = 25.0
    clean = torch.from_numpy(synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1)).cuda().float()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    noisy0 = clean + sigma * torch.randn(clean.shape, generator=g, device="cuda")
    psnr = lambda x: float((20 * torch.log10(255.0 / torch.sqrt(((x - clean) ** 2).mean(dim=1)))).mean())
    ctx = L.Context(0)
    for k in ("LFBM5D_EMULATE_WORLD", "LFBM5D_DATA_DRIVEN_SCHEDULE"):
        os.environ.pop(k, None)
    os.environ["LFBM5D_DATA_DRIVEN_SCHEDULE"] = "1"
    b0, d0, w0 = run(ctx, noisy0, ah, aw, H, W, sigma)
    os.environ.pop("LFBM5D_DATA_DRIVEN_SCHEDULE")
    print(f"sequential, data-driven schedule: windows {len(w0)} psnr basic {psnr(b0):.4f} denoised {psnr(d0):.4f}")
    plan = core.plan_windows(aw, ah, 1, L.ROWMAJOR)
    print("planned sequence == data-driven sequence:", bool(np.array_equal(plan, w0)))
    b1, d1, w1 = run(ctx, noisy0, ah, aw, H, W, sigma)
    print("planned (default), one rank: bit-identical", bool(torch.equal(b0, b1) and torch.equal(d0, d1)))
    for n in (2, 4, 8):
        os.environ["LFBM5D_EMULATE_WORLD"] = str(n)
        b, d, w = run(ctx, noisy0, ah, aw, H, W, sigma)
        print(f"{n} ranks: windows {len(w)} psnr basic {psnr(b):.4f} ({psnr(b) - psnr(b0):+.4f}) denoised {psnr(d):.4f} ({psnr(d) - psnr(d0):+.4f}) "
              f"max abs diff {float((d - d0).abs().max()):.3f}")
    os.environ.pop("LFBM5D_EMULATE_WORLD")


if __name__ == "__main__":
    main()
