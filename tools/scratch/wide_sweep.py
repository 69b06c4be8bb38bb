"""Random 5x5 / 7x7 window passes through the wide-window pass test (development aid).  usage: wide_sweep.py SEED NCASES"""
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
    aw = int(rng.choice([5, 7]))
    step = int(rng.integers(1, 3))
    k = int(rng.choice([8, 8, 12, 16]))
    N = int(rng.choice([1, 2, 4, 8] if step == 1 else [2, 4, 8, 16]))
    nSim, nDisp, p = int(rng.integers(4, 7)), int(rng.integers(1, 3)), int(rng.integers(3, 6))
    t2 = str(rng.choice(["id", "dct", "bior"] if k != 12 else ["id", "dct"])) if step == 1 else str(rng.choice(["dct", "bior"] if k != 12 else ["dct"]))
    t4 = str(rng.choice(["sadct", "dct", "id"]))
    t5 = str(rng.choice(["haar", "hw", "dct"]))
    crop = int(rng.integers(k + 2 * (nSim + nDisp) + 10, 80))
    holes = tuple(int(h) for h in rng.choice(aw * aw, size=int(rng.integers(0, 4)), replace=False) if int(h) != (aw * aw) // 2)
    case = (f"w{ci}", step, (N, nSim, nDisp, k, p, t2, t4, t5), crop, holes)
    try:
        T._wide_window_pass(ctx, case, aw)
        print(aw, case, "ok", flush=True)
    except AssertionError:
        bad += 1
        print(aw, case, "ASSERT", traceback.format_exc().strip().splitlines()[-4:], flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(aw, case, "EXCEPTION", e, flush=True)
print("bad", bad)
