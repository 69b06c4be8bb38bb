import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
import helpers as Hh
from oracle import oracle as O
lf = synth.make_lf(5, 5, 48, 48)
clean, noisy = Hh.noisy_lf(lf, 25.0)
ctx = L.Context(0)
pk = (4, 6, 2, 8, 4, "id", "sadct", "haar")
for name, idx in (("centre", [6,7,8,11,12,13,16,17,18]), ("corner", [12,13,14,17,18,19,22,23,24])):
    win, Wb, Hb = Hh.padded_window(noisy[idx], 48, 48, 3, 8)
    num_o, den_o, st = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3)
    d_win = torch.from_numpy(win).cuda(); d_num = torch.zeros_like(d_win); d_den = torch.zeros_like(d_win)
    ctx.core_pass(1, core.make_params(25.0, 2.7, *pk), 3, 3, Wb, Hb, 3, d_win, None, d_num, d_den, np.ones(9,np.uint32), np.zeros(9,np.uint32), 4, 4)
    ng, dg = d_num.cpu().numpy(), d_den.cpu().numpy()
    eo, eg = Hh.estimate(num_o, den_o, win), Hh.estimate(ng, dg, win)
    print(name, "Wb", Wb, "groups", st.groups, "sadct", st.sadct_groups, "max est diff", np.abs(eo-eg).max(), "den diff", np.abs(den_o-dg).max(), "cov", (den_o>0).mean(), (dg>0).mean())
    d = np.abs(eo-eg).reshape(9,3,Hb,Wb)
    print(" per SAI max diff", d.max(axis=(1,2,3)))
    refs, idx_g, cnt, best, shape = ctx.last_bm(4, 9, Wb*Hb)
    print(" refs", len(refs), "cnt hist", np.bincount(cnt))
