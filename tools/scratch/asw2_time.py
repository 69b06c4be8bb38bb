import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
aw = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
A = aw * aw; sigma = 25.0
lf = synth.make_lf(aw, aw, H, W).reshape(A, 3, H, W).astype(np.float32)
lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
ctx = L.Context(0)
for step, pk in ((1, (8, 18, 6, 16, 4, "id", "sadct", "haar")), (2, (16, 18, 6, 8, 4, "dct", "sadct", "haar"))):
    P = core.make_params(sigma, 2.7, *pk)
    nHW = pk[1] + pk[2]
    pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
    Hb, Wb = pad.shape[2:]
    noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(A, -1)).cuda()
    basic = (0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)) if step == 2 else None
    num = torch.zeros_like(noisy); den = torch.zeros_like(noisy)
    mask = np.ones(A, np.uint32); proc = np.zeros(A, np.uint32)
    reps = 3
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize(); ctx.reset_stats()
        ctx.core_pass(step, P, aw, aw, Wb, Hb, 3, noisy, basic, num, den, mask, proc, A // 2, A // 2)
    torch.cuda.synchronize()
    s = ctx.stats()
    print(f"{aw}x{aw} window, step {step} {Wb}x{Hb}: bm {s.ms_bm/reps:.2f} group {s.ms_group/reps:.2f} agg {s.ms_aggregate/reps:.2f} ms/pass; groups {s.groups//reps} (shape-adaptive {s.sadct_groups//reps})", flush=True)
