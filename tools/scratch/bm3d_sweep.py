"""Random per-SAI BM3D configurations through the BM3D step test (development aid).  usage: bm3d_sweep.py SEED NCASES"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
import lfbm5d_amd as L
rng = np.random.default_rng(int(sys.argv[1]))
ctx = L.Context(0)
bad = 0
for ci in range(int(sys.argv[2])):
    sigma = float(rng.choice([10.0, 25.0, 40.0, 50.0]))
    grey = bool(rng.random() < 0.3)
    k = int(rng.choice([8, 8, 12, 16]))
    n = int(rng.integers(4, 13))
    p = int(rng.integers(2, 6))
    crop = int(rng.integers(k + 2 * n + 8, 104))
    t1 = str(rng.choice(["dct", "bior"] if k != 12 else ["dct"]))
    t2 = str(rng.choice(["dct", "bior"] if k != 12 else ["dct"]))
    N1, N2 = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
    case = (f"rnd{ci}", sigma, grey, crop, (N1, n, k, p, t1, 0), (N2, n, k, p, t2, 0))
    try:
        T.test_bm3d_steps_match_oracle(ctx, case)
        print(case, "ok", flush=True)
    except AssertionError:
        bad += 1
        tb = traceback.format_exc().strip().splitlines()
        print(case, "ASSERT", tb[-3:], flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(case, "EXCEPTION", e, flush=True)
print("bad", bad)
