"""The headline job N times on one context (default lanes): every run's three light fields must be bit-identical to the first's.
usage: python tools/scratch/soak_determinism.py [runs]"""
import hashlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ah = aw = 17; H = W = 512; sigma = 25.0
clean = synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1).astype(np.float32)
noisy = torch.from_numpy(synth.add_noise_mt19937(clean, sigma, seed=1)).cuda()
P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
mask = np.ones(ah * aw, np.uint32)
ctx = L.Context(0)
first = None
for i in range(runs):
    n = noisy.clone(); b = torch.zeros_like(n); d = torch.zeros_like(n)
    t0 = time.perf_counter()
    ctx.denoise(P1, P2, n, mask, b, d, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = tuple(hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12] for x in (n, b, d))
    first = first or h
    print(f"run {i}: {dt * 1e3:.1f} ms  {h}  {'same' if h == first else 'DIFFERENT'}", flush=True)
    assert h == first
print("all", runs, "runs bit-identical")
