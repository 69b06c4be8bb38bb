"""One case of the random parity sweep with its error statistics printed (development aid).
usage: LFBM5D_TEST_SEED=4 LFBM5D_TEST_CASES=16 python tools/scratch/rnd_case.py rnd0-"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
import helpers as Hh
import lfbm5d_amd as L
ctx = L.Context(0)
for case in T._random_cases():
    if not case[0].startswith(sys.argv[1]):
        continue
    name, step, sigma, pk, (ch, cw), useSD = case
    win, Wb, Hb, Cc = T.window(sigma, pk, (ch, cw)) if hasattr(T, "window") else None
    basic = None
    if step == 2:
        n1, d1, _ = Hh.oracle_pass(1, sigma, (pk[0] // 2 or 1,) + pk[1:5] + ("id", "sadct", "haar"), win, None, Wb, Hb, Cc)
        basic = np.ascontiguousarray(Hh.estimate(n1, d1, win).astype(np.float32))
    num_o, den_o, st = Hh.oracle_pass(step, sigma, pk, win, basic, Wb, Hb, Cc, useSD=useSD)
    for env in ("", "generic"):
        if env: os.environ["LFBM5D_GROUP_GENERIC"] = "1"
        num_g, den_g = T.gpu_pass(ctx, step, sigma, pk, win, basic, Wb, Hb, Cc, useSD=useSD)
        print("  non-finite: oracle num %d den %d | gpu num %d den %d | same den positions %s, same num positions %s" % (
            (~np.isfinite(num_o)).sum(), (~np.isfinite(den_o)).sum(), (~np.isfinite(num_g)).sum(), (~np.isfinite(den_g)).sum(),
            np.array_equal(~np.isfinite(den_o), ~np.isfinite(den_g)), np.array_equal(~np.isfinite(num_o), ~np.isfinite(num_g))))
        fin = np.isfinite(den_o) & np.isfinite(den_g) & np.isfinite(num_o) & np.isfinite(num_g)
        den_g, den_o, num_g, num_o = np.where(fin, den_g, 0), np.where(fin, den_o, 0), np.where(fin, num_g, 0), np.where(fin, num_o, 0)
        rel = np.abs(den_g - den_o) / np.maximum(np.abs(den_o), 1e-30)
        rel = rel[den_o != 0]
        eo, eg = Hh.estimate(num_o, den_o, win), Hh.estimate(num_g, den_g, win)
        print(name, env or "default", "rel den: max %.3g q999 %.3g q99 %.3g mean %.3g | bad(1e-4) %.3g | est diff max %.3g mean %.3g"
              % (rel.max(), np.quantile(rel, 0.999), np.quantile(rel, 0.99), rel.mean(), (rel > 1e-4).mean(), np.abs(eo - eg).max(), np.abs(eo - eg).mean()))
