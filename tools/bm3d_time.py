#!/usr/bin/env python3
"""Time of the per-SAI BM3D (LFBM3Ddenoising, README parameters) on a synthetic light field held in HBM.
usage: python tools/bm3d_time.py [n_sai] [H]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    lf = synth.make_lf(3, 3, H, H).reshape(9, -1).astype(np.float32)[np.arange(n) % 9]
    lf = lf + 25.0 * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    ctx = L.Context(0)
    hard = core.make_bm3d_params(25.0, 2.7, 16, 16, 8, 3, "bior")
    wien = core.make_bm3d_params(25.0, 2.7, 32, 16, 8, 3, "dct")
    d_n = torch.from_numpy(lf).cuda()
    d_b = torch.zeros_like(d_n); d_d = torch.zeros_like(d_n)
    mask = np.ones(n, np.uint32)
    for it in range(2):
        ctx.reset_stats()
        torch.cuda.synchronize(); t0 = time.time()
        ctx.bm3d_lf(hard, wien, d_n.clone(), mask, d_b, d_d, H, H, 3)
        torch.cuda.synchronize(); dt = time.time() - t0
    s = ctx.stats()
    print(f"{n} SAIs {H}x{H}: {dt * 1e3:.1f} ms = {n * H * H / 1e6 / dt:.1f} SAI-MP/s; per SAI bm {s.ms_bm / n:.2f} group {s.ms_group / n:.2f} "
          f"agg {s.ms_aggregate / n:.2f} ms; groups {s.groups // n} per SAI")


if __name__ == "__main__":
    main()
