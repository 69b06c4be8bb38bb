"""Group-stage time of the configurations outside the README shapes (the general kernels): one 3x3x256^2 window pass (304^2 padded) each,
and the 5x5 / 7x7 windows of both steps.  usage: python tools/scratch/slow_configs.py [quick]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
H = W = 256; sigma = 25.0
ctx = L.Context(0)
def run(aw, step, pk):
    A = aw * aw
    lf = synth.make_lf(aw, aw, H, W).reshape(A, 3, H, W).astype(np.float32)
    lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    P = core.make_params(sigma, 2.7, *pk)
    nHW = pk[1] + pk[2]
    pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
    Hb, Wb = pad.shape[2:]
    noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(A, -1)).cuda()
    basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
    num = torch.zeros_like(noisy); den = torch.zeros_like(noisy)
    mask = np.ones(A, np.uint32); proc = np.zeros(A, np.uint32)
    reps = 2
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize(); ctx.reset_stats()
        ctx.core_pass(step, P, aw, aw, Wb, Hb, 3, noisy, basic, num, den, mask, proc, A // 2, A // 2)
    torch.cuda.synchronize()
    s = ctx.stats()
    return s.ms_bm / reps, s.ms_group / reps, s.ms_aggregate / reps
CASES = [(3, 1, (8, 18, 6, 12, 4, "dct", "sadct", "haar")), (3, 1, (8, 18, 6, 16, 4, "dct", "sadct", "dct")),
         (3, 2, (16, 18, 6, 8, 4, "dct", "sadct", "haar")), (3, 2, (32, 18, 6, 8, 4, "dct", "sadct", "haar")),
         (3, 2, (16, 18, 6, 12, 4, "dct", "sadct", "haar")), (3, 2, (16, 18, 6, 16, 4, "dct", "sadct", "haar")),
         (3, 2, (32, 18, 6, 16, 4, "dct", "sadct", "dct")), (3, 2, (16, 18, 6, 16, 4, "bior", "sadct", "haar")),
         (3, 2, (16, 18, 6, 16, 4, "id", "sadct", "haar")), (3, 2, (32, 18, 6, 8, 4, "id", "sadct", "haar")),
         (5, 2, (16, 18, 6, 8, 4, "dct", "sadct", "haar")), (7, 2, (16, 18, 6, 8, 4, "dct", "sadct", "haar")),
         (5, 1, (8, 18, 6, 16, 4, "bior", "sadct", "haar")), (5, 1, (8, 18, 6, 16, 4, "id", "sadct", "haar"))]
if len(sys.argv) > 1: CASES = CASES[2:7] + CASES[10:11]
for aw, step, pk in CASES:
    bm, g, a = run(aw, step, pk)
    print(f"{aw}x{aw} step {step} {pk[5]:4s} k{pk[3]:2d} {pk[7]:4s} N{pk[0]:2d}: bm {bm:.2f} group {g:.2f} agg {a:.2f}", flush=True)
