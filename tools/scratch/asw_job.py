"""Whole two-step job with wide angular windows on a synthetic light field: time and PSNR.
usage: python tools/scratch/asw_job.py <angular size> <aswSize> [H]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
a = int(sys.argv[1]); an = int(sys.argv[2]); H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sigma = 25.0
lf = synth.make_lf(a, a, H, W).reshape(a * a, -1).astype(np.float32)
noisy = lf + sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
P1 = core.make_params(sigma, 2.7, 8, 18, 6, 16, 4, "id", "sadct", "haar")
P2 = core.make_params(sigma, 2.7, 16, 18, 6, 8, 4, "dct", "sadct", "haar")
mask = np.ones(a * a, np.uint32)
ctx = L.Context(0)
d_n0 = torch.from_numpy(noisy).cuda()
for it in range(2):
    d_n = d_n0.clone(); d_b = torch.zeros_like(d_n); d_o = torch.zeros_like(d_n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.denoise(P1, P2, d_n, mask, d_b, d_o, L.ROWMAJOR, a, a, an, an, W, H, 3)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
def psnr(x): return 10 * np.log10(255.0 ** 2 / np.mean((np.clip(x, 0, 255) - lf) ** 2))
print(f"{a}x{a}x{H}x{W}, aswSize {an}: {t * 1e3:.1f} ms per job, {a * a * H * W / 1e6 / t:.1f} SAI-MP/s, {len(ctx.last_windows())} windows; PSNR noisy {psnr(noisy):.2f} basic {psnr(d_b.cpu().numpy()):.2f} denoised {psnr(d_o.cpu().numpy()):.2f}", flush=True)
