import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch, helpers as Hh
from oracle import oracle as O
import lfbm5d_amd as L
from lfbm5d_amd import core
ctx = L.Context(0)
for major in ("row", "col"):
    lf = Hh.textured_lf(5, 5, 64, 64)
    if major == "col":
        lf = np.ascontiguousarray(lf.reshape(5, 5, 3, 64, 64).transpose(1, 0, 2, 3, 4)).reshape(25, 3, 64, 64)
    mo, mg = (O.ROWMAJOR, L.ROWMAJOR) if major == "row" else (O.COLMAJOR, L.COLMAJOR)
    clean, noisy = Hh.noisy_lf(lf, 25.0)
    mask = np.ones(25, np.uint32)
    p1, p2 = (4, 6, 2, 8, 4, "id", "sadct", "haar"), (8, 6, 2, 8, 4, "dct", "sadct", "haar")
    for mw in (1, 2, 3, 5):
        n1, b_o, st1 = O.run_step1(O.make_params(25.0, 2.7, *p1), noisy.copy(), mask, mo, 5, 5, 1, 64, 64, 3, max_windows=mw)
        os.environ["LFBM5D_MAX_WINDOWS"] = str(mw)
        d_noisy = torch.from_numpy(noisy).cuda(); d_basic = torch.zeros_like(d_noisy)
        ctx.step1(core.make_params(25.0, 2.7, *p1), d_noisy, mask, d_basic, mg, 5, 5, 1, 64, 64, 3)
        b_g = d_basic.cpu().numpy()
        d = np.abs(b_g - b_o)
        print(major, "windows", mw, "max|diff|", d.max(), "per-SAI max", np.round(d.max(axis=1), 4).tolist(), st1.sadct_groups)
        if mw == 2:
            print("   frac |d|>1e-3:", float((d > 1e-3).mean()), " >0.1:", float((d > 0.1).mean()), " >1:", float((d > 1).mean()))
