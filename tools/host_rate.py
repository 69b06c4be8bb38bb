#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline job: lfbm5d_denoise_host (the drop-in boundary's host-buffer form: pageable numpy arrays
in, both outputs back; round 5: streamed through the window graph, LFBM5D_HOST_BLOCKING=1 = rounds 1-4's four blocking copies) and
the C++ drop-in on vector<vector<float>> light fields (run_bm5d_1st_step + run_bm5d_2nd_step = the interval the reference times,
and run_bm5d) against lfbm5d_denoise_device on the same light field.  bench.py reports the same numbers under `seam`; its
`value` is always the device-resident one.
usage: python tools/host_rate.py [reps] [ah aw H W]"""
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
    if len(sys.argv) > 5:
        ah, aw, H, W = (int(v) for v in sys.argv[2:6])
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
    for blocking in (False, True):
        if blocking:
            os.environ["LFBM5D_HOST_BLOCKING"] = "1"
        for it in range(reps + 1):
            src = noisy_h.copy()
            t0 = time.perf_counter()
            ctx.denoise(P1, P2, src, mask, basic_h, out_h, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
            t = time.perf_counter() - t0
            if it:
                print(f"host buffers ({'blocking copies' if blocking else 'streamed'}): {t * 1e3:8.1f} ms  {mpix / t:7.1f} SAI-MP/s  (pageable, 1 light field in, 3 out)")
        print("outputs identical:", bool(np.array_equal(ref_o, out_h)))
        os.environ.pop("LFBM5D_HOST_BLOCKING", None)
    # the C++ drop-in on vector<vector<float>> light fields
    hard, wien = (8, 18, 6, 16, 4, "id", "sadct", "haar"), (16, 18, 6, 8, 4, "dct", "sadct", "haar")
    for one_job in (False, True):
        ms, n, b, d = core.dropin_probe(noisy_h, mask, aw, ah, W, H, 3, sigma, 2.7, hard, wien, one_job=one_job, reps=reps + 1)
        for r in range(1, reps + 1):
            t = (ms[r, 0] + ms[r, 1]) * 1e-3
            print(f"drop-in vectors ({'run_bm5d' if one_job else 'run_bm5d_1st_step + run_bm5d_2nd_step'}): {ms[r, 0]:8.1f} + {ms[r, 1]:8.1f} ms  {mpix / t:7.1f} SAI-MP/s")
        print("outputs identical:", bool(np.array_equal(ref_o, d)))


if __name__ == "__main__":
    main()
