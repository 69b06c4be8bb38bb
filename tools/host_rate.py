#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline job: lfbm5d_denoise_host (the drop-in boundary's host-buffer form: pageable numpy arrays
in, both outputs back) against lfbm5d_denoise_device on the same light field.  DESIGN.md section 5 quotes the result; bench.py's
`value` is always the device-resident one.
usage: python tools/host_rate.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ah = aw = 17
    H = W = 512
    sigma = 25.0
    lf = synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1).astype(np.float32)
    noisy_h = lf + sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    del lf
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    mask = np.ones(ah * aw, np.uint32)
    ctx = L.Context(0)
    mpix = ah * aw * H * W / 1e6
    # device-resident
    d_n0 = torch.from_numpy(noisy_h).cuda()
    d_n, d_b, d_o = torch.empty_like(d_n0), torch.zeros_like(d_n0), torch.zeros_like(d_n0)
    for it in range(reps + 1):
        d_n.copy_(d_n0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.denoise(P1, P2, d_n, mask, d_b, d_o, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if it:
            print(f"device buffers: {t * 1e3:8.1f} ms  {mpix / t:7.1f} SAI-MP/s")
    ref_o = d_o.cpu().numpy()
    del d_n0, d_n, d_b, d_o
    torch.cuda.empty_cache()
    # host buffers (pageable)
    basic_h, out_h = np.zeros_like(noisy_h), np.zeros_like(noisy_h)
    for it in range(reps + 1):
        src = noisy_h.copy()
        t0 = time.perf_counter()
        ctx.denoise(P1, P2, src, mask, basic_h, out_h, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
        t = time.perf_counter() - t0
        if it:
            print(f"host buffers:   {t * 1e3:8.1f} ms  {mpix / t:7.1f} SAI-MP/s  (0.91 GB in, 2 x 0.91 GB out, pageable)")
    print("outputs identical:", bool(np.array_equal(ref_o, out_h)))


if __name__ == "__main__":
    main()
