"""Quick on-GPU parity/timing probe (development aid; the judged tests live in tests/)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lfbm5d_amd as L
from lfbm5d_amd import core
from oracle import oracle as O


def padded_window(lf_rgb_u8, sigma, nHW, cs=O.OPP, crop=None):
    """lf: [9][3][H][W] uint8 -> noisy (colour-transformed, padded) window arrays [9][3*Hb*Wb]."""
    lf = lf_rgb_u8.astype(np.float32)
    if crop:
        lf = lf[:, :, :crop, :crop]
    A, Cc, H, W = lf.shape
    lf = np.ascontiguousarray(lf).reshape(A, -1)
    noisy = O.add_noise_lf(lf, sigma, seed=1)
    lib = O.lib()
    Wb, Hb = W + 2 * nHW, H + 2 * nHW
    out = np.zeros((A, Cc * Wb * Hb), np.float32)
    for st in range(A):
        im = noisy[st].copy()
        lib.orc_color_transform(im, cs, W, H, Cc, 1)
        lib.orc_symetrize(im, out[st], W, H, Cc, nHW)
    return out, Wb, Hb, Cc


def check_pass(ctx, name, lf, sigma, step, pk, crop=None, basic_from=None):
    P_or = O.make_params(sigma, 2.7, *pk)
    P_gp = core.make_params(sigma, 2.7, *pk)
    N, nSim, nDisp, k, p = pk[:5]
    nHW = nSim + nDisp
    noisy, Wb, Hb, Cc = padded_window(lf, sigma, nHW, crop=crop)
    A = 9
    plane = Wb * Hb
    basic = None
    if step == 2:
        basic = noisy * 0.5 + np.roll(noisy, 1, axis=1) * 0.5  # any smooth-ish pilot works for parity
    mask = np.ones(A, np.uint32)
    proc = np.zeros(A, np.uint32)
    num_o = np.zeros_like(noisy)
    den_o = np.zeros_like(noisy)
    st = O.Stats()
    t0 = time.time()
    rc = O.lib().orc_pass(step, C.byref(P_or), 3, 3, Wb, Hb, Cc, noisy.reshape(-1),
                          basic.ctypes.data if basic is not None else None, num_o.reshape(-1), den_o.reshape(-1),
                          mask, proc, 4, 4, 0, -1, C.byref(st))
    t_or = time.time() - t0
    assert rc == 0
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic = torch.from_numpy(basic).cuda() if basic is not None else None
    d_num = torch.zeros_like(d_noisy)
    d_den = torch.zeros_like(d_noisy)
    torch.cuda.synchronize()
    ctx.reset_stats()
    t0 = time.time()
    ctx.core_pass(step, P_gp, 3, 3, Wb, Hb, Cc, d_noisy, d_basic, d_num, d_den, mask, proc, 4, 4)
    t_gp = time.time() - t0
    s = ctx.stats()
    num_g = d_num.cpu().numpy()
    den_g = d_den.cpu().numpy()
    # BM tables vs oracle
    refs, idx, cnt, best, shape = ctx.last_bm(N, A, plane)
    est = noisy[:, :plane] if step == 1 else basic[:, :plane]
    sig = np.zeros(3, np.float32)
    O.lib().orc_sigma_table(sigma, Cc, O.OPP, sig)
    tau = (3.0 if Cc == 1 else 1.0) * ((3000 if step == 1 else 2000) if sig[0] < 35 else 5000)
    o_idx = np.zeros((len(refs), max(N, 1)), np.uint32)
    o_cnt = np.zeros(len(refs), np.uint32)
    O.lib().orc_bm_self(np.ascontiguousarray(est[4]), Wb, Hb, k, N, nHW, nSim, tau, refs, len(refs), o_idx.reshape(-1), o_cnt)
    same_cnt = (o_cnt == cnt).mean()
    same_idx = np.mean([np.array_equal(o_idx[r, :o_cnt[r]], idx[r, :cnt[r]]) for r in range(len(refs))])
    ob = np.zeros(plane, np.uint32)
    osh = np.zeros(plane, np.uint8)
    O.lib().orc_bm_stereo(np.ascontiguousarray(est[4]), np.ascontiguousarray(est[7]), Wb, Hb, k, nDisp, tau, ob, osh)
    yy = slice(nDisp, Hb - k - nDisp + 1)
    b_o = ob.reshape(Hb, Wb)[yy, yy]
    b_g = best[7].reshape(Hb, Wb)[yy, yy]
    s_o = osh.reshape(Hb, Wb)[yy, yy]
    s_g = shape[7].reshape(Hb, Wb)[yy, yy]
    print(f"[{name}] refs {len(refs)} self cnt match {same_cnt:.6f} idx match {same_idx:.6f} "
          f"stereo best match {(b_o == b_g).mean():.6f} shape match {(s_o == s_g).mean():.6f}")
    if not (b_o == b_g).all():
        bad = np.argwhere(b_o != b_g)
        print(f"[{name}] stereo mismatches {len(bad)}: rows {bad[:, 0].min() + nDisp}..{bad[:, 0].max() + nDisp} "
              f"cols {bad[:, 1].min() + nDisp}..{bad[:, 1].max() + nDisp} (window {Wb}x{Hb}); first {bad[:6].tolist()}")
    dn = np.abs(num_g - num_o).max() / max(1e-9, np.abs(num_o).max())
    dd = np.abs(den_g - den_o).max() / max(1e-9, np.abs(den_o).max())
    cov = ((den_o > 0) == (den_g > 0)).mean()
    eo = np.where(den_o > 0, num_o / np.where(den_o > 0, den_o, 1), 0)
    eg = np.where(den_g > 0, num_g / np.where(den_g > 0, den_g, 1), 0)
    print(f"[{name}] oracle {t_or:.2f}s gpu {t_gp * 1e3:.1f}ms (bm {s.ms_bm:.2f} group {s.ms_group:.2f} agg {s.ms_aggregate:.2f} ms) "
          f"groups {s.groups}/{st.groups} stack {s.stack_patches}/{st.stack_patches} sadct {s.sadct_groups}/{st.sadct_groups}")
    print(f"[{name}] rel max |num| diff {dn:.3e} |den| diff {dd:.3e} coverage agree {cov:.6f} "
          f"estimate max abs diff {np.abs(eo - eg).max():.4e} mean {np.abs(eo - eg).mean():.3e}")
    sys.stdout.flush()


def check_e2e(ctx, name, lf, sigma, p1, p2):
    A, Cc, H, W = lf.shape
    clean = np.ascontiguousarray(lf.astype(np.float32)).reshape(A, -1)
    noisy = O.add_noise_lf(clean, sigma, seed=1)
    mask = np.ones(A, np.uint32)
    t0 = time.time()
    n1, b_o, _ = O.run_step1(O.make_params(sigma, 2.7, *p1), noisy.copy(), mask, O.ROWMAJOR, 3, 3, 1, W, H, Cc)
    n2, b2, d_o, _ = O.run_step2(O.make_params(sigma, 2.7, *p2), n1, b_o.copy(), mask, O.ROWMAJOR, 3, 3, 1, W, H, Cc)
    t_or = time.time() - t0
    d_noisy = torch.from_numpy(noisy).cuda()
    d_basic = torch.zeros_like(d_noisy)
    d_den = torch.zeros_like(d_noisy)
    torch.cuda.synchronize()
    for it in range(2):
        d_noisy.copy_(torch.from_numpy(noisy))
        torch.cuda.synchronize()
        ctx.reset_stats()
        t0 = time.time()
        ctx.step1(core.make_params(sigma, 2.7, *p1), d_noisy, mask, d_basic, L.ROWMAJOR, 3, 3, 1, W, H, Cc)
        t1 = time.time()
        basic_g = d_basic.cpu().numpy()
        ctx.step2(core.make_params(sigma, 2.7, *p2), d_noisy, mask, d_basic, d_den, L.ROWMAJOR, 3, 3, 1, W, H, Cc)
        t2 = time.time()
    s = ctx.stats()
    den_g = d_den.cpu().numpy()
    print(f"[{name}] oracle {t_or:.1f}s | gpu step1 {1e3 * (t1 - t0):.1f} ms step2 {1e3 * (t2 - t1):.1f} ms "
          f"(bm {s.ms_bm:.2f} group {s.ms_group:.2f} agg {s.ms_aggregate:.2f})")
    print(f"[{name}] PSNR basic oracle {O.psnr_lf(b_o, clean):.6f} gpu {O.psnr_lf(basic_g, clean):.6f} | "
          f"denoised oracle {O.psnr_lf(d_o, clean):.6f} gpu {O.psnr_lf(den_g, clean):.6f} | "
          f"max abs diff basic {np.abs(basic_g - b_o).max():.4f} denoised {np.abs(den_g - d_o).max():.4f}")
    sys.stdout.flush()


if __name__ == "__main__":
    lf = np.load("tests/golden/sourceLF_3x3_256_u8.npy")
    ctx = L.Context(0)
    which = sys.argv[1:] or ["small", "readme", "e2e"]
    if "small" in which:
        check_pass(ctx, "small-ht-id", lf, 25.0, 1, (4, 6, 2, 8, 3, "id", "sadct", "haar"), crop=64)
        check_pass(ctx, "small-ht-bior", lf, 25.0, 1, (4, 6, 2, 8, 3, "bior", "sadct", "haar"), crop=64)
        check_pass(ctx, "small-wien-dct", lf, 25.0, 2, (8, 6, 2, 8, 3, "dct", "sadct", "haar"), crop=64)
        check_pass(ctx, "small-ht-n1", lf, 50.0, 1, (1, 6, 2, 16, 3, "bior", "sadct", "haar"), crop=96)
        check_pass(ctx, "small-ht-hw", lf, 25.0, 1, (4, 6, 2, 8, 3, "dct", "dct", "hw"), crop=64)
    if "k16n8" in which:
        check_pass(ctx, "ht-k16-n8", lf, 25.0, 1, (8, 8, 3, 16, 4, "id", "sadct", "haar"), crop=96)
    if "c4" in which:
        check_pass(ctx, "c4-ht-bior", lf, 10.0, 1, (8, 18, 6, 16, 4, "bior", "sadct", "haar"))
    if "readme" in which:
        check_pass(ctx, "readme-ht", lf, 25.0, 1, (8, 18, 6, 16, 4, "id", "sadct", "haar"))
        check_pass(ctx, "readme-wien", lf, 25.0, 2, (16, 18, 6, 8, 4, "dct", "sadct", "haar"))
    if "e2e" in which:
        check_e2e(ctx, "e2e-readme", lf, 25.0, (8, 18, 6, 16, 4, "id", "sadct", "haar"), (16, 18, 6, 8, 4, "dct", "sadct", "haar"))
        check_e2e(ctx, "e2e-c4", lf, 10.0, (8, 18, 6, 16, 4, "bior", "sadct", "haar"), (16, 18, 6, 8, 4, "dct", "sadct", "haar"))
        check_e2e(ctx, "e2e-c5", lf, 50.0, (1, 18, 3, 16, 3, "bior", "sadct", "haar"), (8, 18, 3, 8, 3, "dct", "sadct", "haar"))
