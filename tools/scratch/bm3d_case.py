"""Difference statistics of one BM3D hard-threshold step, GPU against the oracle (development aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from oracle import oracle as O
import lfbm5d_amd as L
from lfbm5d_amd import core
ctx = L.Context(0)
for (sigma, grey, crop, hard) in ((25.0, False, 101, (8, 4, 12, 3, 'dct', 0)), (25.0, False, 94, (4, 5, 12, 4, 'dct', 0)), (10.0, True, 56, (32, 6, 12, 2, 'dct', 0))):
    lf = Hh.source_lf(crop=crop)[:1]
    if grey: lf = lf[:, :1]
    Cc = lf.shape[1]
    clean, noisy = Hh.noisy_lf(lf, sigma)
    nP = hard[1]
    win, Wb, Hb = Hh.padded_window(noisy, crop, crop, Cc, nP)
    b_o, st1 = O.bm3d_step(1, sigma, 2.7, win[0], None, Wb, Hb, Cc, hard[1], hard[2], hard[0], hard[3], hard[4], hard[5])
    d_win = torch.from_numpy(win[0]).cuda(); d_out = torch.zeros_like(d_win)
    ctx.reset_stats()
    ctx.bm3d_step(1, core.make_bm3d_params(sigma, 2.7, *hard), Wb, Hb, Cc, d_win, None, d_out)
    s = ctx.stats()
    d = np.abs(d_out.cpu().numpy() - b_o).reshape(Cc, Hb, Wb)[:, nP:-nP, nP:-nP]
    print((sigma, grey, crop, hard), "groups", s.groups, st1.groups, "patches", s.stack_patches, st1.stack_patches,
          "max %.4g  > 2e-3: %d px of %d, mean %.3g; per channel max" % (d.max(), (d > 2e-3).sum(), d.size, d.mean()), [float(x.max()) for x in d])
