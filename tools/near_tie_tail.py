#!/usr/bin/env python3
"""How the GPU's and the CPU oracle's results drift apart over the windows of a step (round-5 review, item 6): near-tie
hard-threshold decisions (float32 transforms here, double accumulation in the oracle) flip a coefficient now and then, every
flip changes one group's weight and one filtered patch stack, and later windows match blocks on the running estimate.

The first N windows of each step of the headline workload (17x17x512x512, sigma 25, MT19937 noise seed 1, README parameters)
through the oracle (orc_run_step*, max_windows = N) and through the C-ABI (option max_windows = N), then per WINDOW INDEX w the
SAIs whose last toucher is window w: pixels off by more than 1 / 0.1 grey level, mean and maximum difference, PSNR of either
against the clean light field.  Step 2 starts, on both sides, from the GPU's full step-1 outputs, so its rows show the Wiener
step's own drift.  About 8 s of CPU per window pass on 128 cores.
usage: python tools/near_tie_tail.py [N = 10] > profiles/<tag>_near_tie_tail.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402 -- the checker


def window_sais(pst, aw, ah):
    ps, pt = pst // aw, pst % aw
    s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
    return [(s0 + s) * aw + (t0 + t) for s in range(3) for t in range(3)]


def report(tag, wins, cpu_in, cpu_est, gpu_est, clean, aw, ah):
    last = {}
    for w, pst in enumerate(wins):
        for st in window_sais(int(pst), aw, ah):
            last[st] = w
    print(f"## {tag}: {len(wins)} windows, {len(last)} SAIs touched")
    print("window  SAIs  pixels      >1.0    >0.1   mean|d|    max|d|   psnr_cpu  psnr_gpu  delta_dB")
    tot = np.zeros(3, np.int64)
    for w in range(len(wins)):
        sais = sorted(st for st, lw in last.items() if lw == w)
        if not sais:
            continue
        d = np.abs(cpu_est[sais].astype(np.float64) - gpu_est[sais])
        mse_c = ((cpu_est[sais].astype(np.float64) - clean[sais]) ** 2).mean(axis=1)
        mse_g = ((gpu_est[sais].astype(np.float64) - clean[sais]) ** 2).mean(axis=1)
        pc, pg = float((20 * np.log10(255.0 / np.sqrt(mse_c))).mean()), float((20 * np.log10(255.0 / np.sqrt(mse_g))).mean())
        tot += (d.size, int((d > 1.0).sum()), int((d > 0.1).sum()))
        print(f"{w:6d} {len(sais):5d} {d.size:9d} {int((d > 1.0).sum()):7d} {int((d > 0.1).sum()):7d} {d.mean():9.2e} {d.max():9.2e} {pc:9.4f} {pg:9.4f} {pg - pc:+9.1e}")
    print(f"total  {len(last):5d} {tot[0]:9d} {tot[1]:7d} {tot[2]:7d}")


def main():
    n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ah = aw = 17
    H = W = 512
    sigma = 25.0
    p1, p2 = (8, 18, 6, 16, 4, "id", "sadct", "haar"), (16, 18, 6, 8, 4, "dct", "sadct", "haar")
    clean = synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1).astype(np.float32)
    noisy = synth.add_noise_mt19937(clean, sigma, seed=1)
    mask = np.ones(ah * aw, np.uint32)
    ctx = L.Context(0)
    print(f"# tools/near_tie_tail.py {n_win}: headline workload, first {n_win} windows of each step, GPU (C-ABI) against the CPU oracle; "
          f"oracle threads {int(O.lib().orc_get_threads())}")
    # ---- step 1
    t0 = time.time()
    n_o, b_o, st_o = O.run_step1(O.make_params(sigma, 2.7, *p1), noisy.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3, max_windows=n_win)
    w_o = O.last_windows()
    t_cpu = time.time() - t0
    ctx.set_option("max_windows", n_win)
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic = torch.zeros_like(d_noisy)
    ctx.step1(core.make_params(sigma, 2.7, *p1), d_noisy, mask, d_basic, L.ROWMAJOR, aw, ah, 1, W, H, 3)
    w_g = ctx.last_windows()
    assert np.array_equal(np.asarray(w_o)[:len(w_g)], w_g), "window sequences differ"
    print(f"# step 1: oracle {t_cpu:.1f} s for {len(w_o)} windows")
    report("step 1 (hard thresholding), basic estimate", w_g, n_o, b_o, d_basic.cpu().numpy(), clean, aw, ah)
    # ---- step 2 from the GPU's FULL step-1 outputs on both sides
    ctx.set_option("max_windows", None)
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic = torch.zeros_like(d_noisy)
    ctx.step1(core.make_params(sigma, 2.7, *p1), d_noisy, mask, d_basic, L.ROWMAJOR, aw, ah, 1, W, H, 3)
    n1, b1 = d_noisy.cpu().numpy(), d_basic.cpu().numpy()
    t0 = time.time()
    _, bs_o, d_o, _ = O.run_step2(O.make_params(sigma, 2.7, *p2), n1.copy(), b1.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3, max_windows=n_win)
    t_cpu = time.time() - t0
    ctx.set_option("max_windows", n_win)
    d_den = torch.zeros_like(d_noisy)
    ctx.step2(core.make_params(sigma, 2.7, *p2), d_noisy, mask, d_basic, d_den, L.ROWMAJOR, aw, ah, 1, W, H, 3)
    w_g2 = ctx.last_windows()
    print(f"# step 2: oracle {t_cpu:.1f} s for {n_win} windows")
    report("step 2 (Wiener), denoised estimate, both sides from the GPU's full basic estimate", w_g2, bs_o, d_o, d_den.cpu().numpy(), clean, aw, ah)
    ctx.close()


if __name__ == "__main__":
    main()
