import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
ah = aw = 9; H = W = 512; sigma = 25.0
lf = synth.make_lf(ah, aw, H, W).reshape(ah * aw, 3, H * W).astype(np.float32)
for C in (3, 1):
    src = np.ascontiguousarray(lf[:, :C].reshape(ah * aw, -1))
    noisy_h = src + sigma * np.random.default_rng(1).standard_normal(src.shape).astype(np.float32)
    P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
    P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
    ctx = L.Context(0)
    mask = np.ones(ah * aw, np.uint32)
    d_n0 = torch.from_numpy(noisy_h).cuda()
    d_n, d_b, d_o = torch.empty_like(d_n0), torch.zeros_like(d_n0), torch.zeros_like(d_n0)
    for it in range(3):
        d_n.copy_(d_n0); torch.cuda.synchronize(); ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.denoise(P1, P2, d_n, mask, d_b, d_o, L.ROWMAJOR, aw, ah, 1, 1, W, H, C)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    s = ctx.stats()
    print(f"C={C}: {t*1e3:.1f} ms windows {s.windows} passes {s.passes} bm {s.ms_bm:.1f} group {s.ms_group:.1f} agg {s.ms_aggregate:.1f} other {s.ms_other:.1f}", flush=True)
    ctx.close()
