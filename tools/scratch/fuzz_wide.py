"""Seeded random sweep of single window passes on wide angular windows against the oracle (the checks of tests/test_gpu_parity.py's
_wide_window_pass): window side, patch size, group size, transforms, empty SAIs -- the wide-window and slab kernels' parameter space.
(A hard-thresholding case may miss the strict bar on the estimate because a coefficient within float round-off of the threshold decides
differently in the double-precision oracle -- the one such case of seed 1, 11x11 HT dct k 16 hw, fails the same way on round 4's general kernel.)
usage: python tools/scratch/fuzz_wide.py [cases] [seed]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the library: one HIP runtime)
import lfbm5d_amd as L
import test_gpu_parity as T
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = L.Context(0)
bad = 0
for i in range(n_cases):
    aw = int(rng.choice([5, 5, 7, 7, 9, 11]))
    step = int(rng.choice([1, 2]))
    tau2 = str(rng.choice(["id", "dct", "bior"]))
    k = int(rng.choice([8, 16] if tau2 == "bior" else ([6, 8, 10, 12, 16] if tau2 == "id" else [8, 12, 16])))
    N = int(rng.choice([2, 4, 8] if aw >= 9 else [2, 4, 8, 16]))
    tau4 = str(rng.choice(["dct", "sadct", "sadct"])); tau5 = str(rng.choice(["haar", "hw", "dct"]))
    nSim, nDisp, p = int(rng.integers(3, 6)), int(rng.integers(1, 3)), int(rng.integers(3, 6))
    crop = int(k + 2 * (nSim + nDisp) + rng.integers(10, 24))
    A = aw * aw
    holes = tuple(int(h) for h in rng.choice([h for h in range(A) if h != A // 2], size=int(rng.integers(0, 4)), replace=False))
    case = (f"fz{i}", step, (N, nSim, nDisp, k, p, tau2, tau4, tau5), crop, holes)
    try:
        T._wide_window_pass(ctx, case, aw)
        print(f"ok   {aw}x{aw} step {step} {case[2]} crop {crop} holes {holes}", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"FAIL {aw}x{aw} step {step} {case[2]} crop {crop} holes {holes}: {str(e)[:200]}", flush=True)
print(f"{n_cases - bad} of {n_cases} cases within the bars")
