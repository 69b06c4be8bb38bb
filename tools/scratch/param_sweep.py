import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
H = W = 256; sigma = 25.0
lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
ctx = L.Context(0)
def run(step, pk):
    P = core.make_params(sigma, 2.7, *pk)
    nHW = pk[1] + pk[2]
    pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
    Hb, Wb = pad.shape[2:]
    noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
    basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
    num = torch.zeros_like(noisy); den = torch.zeros_like(noisy)
    mask = np.ones(9, np.uint32); proc = np.zeros(9, np.uint32)
    reps = 3
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize(); ctx.reset_stats()
        ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic, num, den, mask, proc, 4, 4)
    torch.cuda.synchronize()
    s = ctx.stats()
    return s.ms_bm / reps, s.ms_group / reps, s.ms_aggregate / reps
for step in (1, 2):
    for t2 in (("id", "dct", "bior") if step == 1 else ("dct", "bior", "id")):
        for k in (8, 12, 16):
            if t2 == "bior" and k == 12: continue
            for t5 in ("haar", "hw", "dct"):
                for N in ((8, 1) if step == 1 else (16, 32)):
                    pk = (N, 18, 6, k, 4, t2, "sadct", t5)
                    try:
                        bm, g, a = run(step, pk)
                        flag = "  <-- slow" if g > 1.5 else ""
                        print(f"step {step} {t2:4s} k{k:2d} {t5:4s} N{N:2d}: bm {bm:.2f} group {g:.2f} agg {a:.2f}{flag}", flush=True)
                    except Exception as e:
                        print(f"step {step} {t2} k{k} {t5} N{N}: {str(e)[:80]}", flush=True)
