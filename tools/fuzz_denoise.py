#!/usr/bin/env python3
"""Seeded random sweep of the two-step job (lfbm5d_denoise_device) against the two calls it replaces: random angular sizes, image
sizes, angular search windows per step, empty SAIs, colour spaces, angular orders, parameter sets, lanes, emulated ranks and
window limits.  The bar is the tests' (tests/test_gpu_denoise.py): LF_noisy, basic and denoised bit-identical, same window list,
messages = the plan's.  GPU only, no oracle.   usage: python tools/fuzz_denoise.py [cases] [seed] [largest angular size, default 9]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core  # noqa: E402

ENV = ("LFBM5D_EMULATE_WORLD", "LFBM5D_LANES", "LFBM5D_MAX_WINDOWS", "LFBM5D_FUSED", "LFBM5D_STEP_SHARDING", "LFBM5D_DATA_DRIVEN_SCHEDULE")
HT = [(4, 6, 2, 8, 4, "id", "sadct", "haar"), (4, 6, 2, 8, 4, "id", "dct", "haar"), (2, 5, 2, 8, 4, "dct", "sadct", "haar"),
      (4, 6, 2, 16, 4, "bior", "sadct", "haar"), (8, 5, 3, 8, 3, "id", "sadct", "hw"), (1, 4, 2, 8, 4, "dct", "sadct", "haar")]
WI = [(8, 6, 2, 8, 4, "dct", "sadct", "haar"), (8, 6, 2, 8, 4, "dct", "dct", "haar"), (4, 5, 2, 8, 3, "dct", "sadct", "hw"),
      (8, 4, 3, 8, 4, "bior", "sadct", "haar"), (16, 6, 2, 8, 4, "dct", "sadct", "haar"), (1, 4, 2, 8, 4, "dct", "sadct", "haar")]


def run(ctx, fused, P1, P2, noisy, mask, aw, ah, an, W, H, mj):
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic, d_den = torch.zeros_like(d_noisy), torch.zeros_like(d_noisy)
    ctx.reset_stats()
    if fused:
        ctx.denoise(P1, P2, d_noisy, mask, d_basic, d_den, mj, aw, ah, an[0], an[1], W, H, 3)
        w = ctx.last_windows()
    else:
        ctx.step1(P1, d_noisy, mask, d_basic, mj, aw, ah, an[0], W, H, 3)
        w1 = ctx.last_windows()
        ctx.step2(P2, d_noisy, mask, d_basic, d_den, mj, aw, ah, an[1], W, H, 3)
        w = np.concatenate([w1, ctx.last_windows()])
    return d_noisy.cpu().numpy(), d_basic.cpu().numpy(), d_den.cpu().numpy(), w, ctx.stats()


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    amax = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    rng = np.random.default_rng(seed)
    ctx = L.Context(0)
    bad = 0
    for case in range(n_cases):
        ah, aw = int(rng.integers(3, amax + 1)), int(rng.integers(3, amax + 1))
        an = tuple(int(rng.integers(1, 3)) if min(ah, aw) >= 5 else 1 for _ in range(2))
        H, W = int(rng.integers(40, 73)), int(rng.integers(40, 73))
        pk1, pk2 = HT[int(rng.integers(len(HT)))], WI[int(rng.integers(len(WI)))]
        if pk1[3] == 16:
            H, W = max(H, 56), max(W, 56)
        cs = ("opp", "yuv", "rgb")[int(rng.integers(3))]
        mj = (L.ROWMAJOR, L.COLMAJOR)[int(rng.integers(2))]
        mask = np.ones(ah * aw, np.uint32)
        n_holes = int(rng.integers(0, max(1, ah * aw // 5))) if rng.random() < 0.5 else 0
        mask[rng.choice(ah * aw, n_holes, replace=False)] = 0
        lanes = str(int(rng.integers(1, 4)))
        emu = (None, None, "2", "3", "5", "8")[int(rng.integers(6))]
        maxw = str(int(rng.integers(2, 9))) if rng.random() < 0.2 else None
        desc = f"case {case}: {ah}x{aw}x{H}x{W} an {an} holes {n_holes} {cs} {'row' if mj == L.ROWMAJOR else 'col'} ht {pk1} wi {pk2} lanes {lanes} emu {emu} maxw {maxw}"
        lf = Hh.textured_lf(ah, aw, H, W)
        clean, noisy = Hh.noisy_lf(lf, 25.0, seed=case + 1)
        P1 = core.make_params(25.0, 2.7, *pk1, color_space=cs)
        P2 = core.make_params(25.0, 2.7, *pk2, color_space=cs)
        for k in ENV:
            os.environ.pop(k, None)
        if maxw:
            os.environ["LFBM5D_MAX_WINDOWS"] = maxw
        os.environ["LFBM5D_LANES"] = "1"
        try:
            n0, b0, d0, w0, _ = run(ctx, False, P1, P2, noisy, mask, aw, ah, an, W, H, mj)
            os.environ["LFBM5D_LANES"] = lanes
            if emu:
                os.environ["LFBM5D_EMULATE_WORLD"] = emu
            n1, b1, d1, w1, s1 = run(ctx, True, P1, P2, noisy, mask, aw, ah, an, W, H, mj)
            ok = np.array_equal(w1, w0) and np.array_equal(n1, n0) and np.array_equal(b1, b0) and np.array_equal(d1, d0)
            note = ""
            if emu and not maxw:
                nodes, msgs, info = core.plan_job(aw, ah, int(emu), 1, an=an, mask=mask, ang_major=mj)
                ok = ok and s1.messages == len(msgs)
                note = f"messages {s1.messages} ranks {len(set(nodes[:, 3].tolist()))}"
            print(("ok   " if ok else "FAIL ") + desc + f" windows {len(w0)} {note}", flush=True)
            bad += 0 if ok else 1
        except Exception as e:  # noqa: BLE001
            print("ERR  " + desc + f": {e}", flush=True)
            bad += 1
    print(f"{n_cases - bad} of {n_cases} cases identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
