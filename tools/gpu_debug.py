import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
import lfbm5d_amd as L
from lfbm5d_amd import core
from oracle import oracle as O
lf = np.load("tests/golden/sourceLF_3x3_256_u8.npy")
A, Cc, H, W = lf.shape
sigma = 50.0
p1 = (1, 18, 3, 16, 3, "bior", "sadct", "haar"); p2 = (8, 18, 3, 8, 3, "dct", "sadct", "haar")
clean = np.ascontiguousarray(lf.astype(np.float32)).reshape(A, -1)
noisy = O.add_noise_lf(clean, sigma, seed=1)
mask = np.ones(A, np.uint32)
n1, b_o, _ = O.run_step1(O.make_params(sigma, 2.7, *p1), noisy.copy(), mask, O.ROWMAJOR, 3, 3, 1, W, H, Cc)
n2, b2, d_o, _ = O.run_step2(O.make_params(sigma, 2.7, *p2), n1.copy(), b_o.copy(), mask, O.ROWMAJOR, 3, 3, 1, W, H, Cc)
ctx = L.Context(0)
# GPU step 2 fed with the ORACLE's step-1 outputs: isolates step 2
d_noisy = torch.from_numpy(n1).cuda(); d_basic = torch.from_numpy(b_o).cuda(); d_den = torch.zeros_like(d_noisy)
ctx.step2(core.make_params(sigma, 2.7, *p2), d_noisy, mask, d_basic, d_den, L.ROWMAJOR, 3, 3, 1, W, H, Cc)
g = d_den.cpu().numpy()
diff = np.abs(g - d_o).reshape(A, Cc, H, W)
print("step2 on identical inputs: max abs diff", diff.max(), "psnr", O.psnr_lf(g, clean), O.psnr_lf(d_o, clean))
idx = np.argsort(diff.reshape(-1))[::-1][:15]
for i in idx:
    st, c, y, x = np.unravel_index(i, diff.shape)
    print(st, c, y, x, "oracle", d_o.reshape(A, Cc, H, W)[st, c, y, x], "gpu", g.reshape(A, Cc, H, W)[st, c, y, x], "basic", b_o.reshape(A,Cc,H,W)[st,c,y,x], "clean", clean.reshape(A,Cc,H,W)[st,c,y,x])
print("count diff>1:", (diff > 1).sum(), "of", diff.size, " nan gpu:", np.isnan(g).sum(), "nan or:", np.isnan(d_o).sum())
