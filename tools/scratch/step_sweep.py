"""Random whole-step configurations, GPU against the oracle (development aid; prints, does not assert).
usage: python tools/scratch/step_sweep.py SEED NCASES"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from oracle import oracle as O
import lfbm5d_amd as L
from lfbm5d_amd import core

seed, ncases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = L.Context(0)
worst = 0.0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for ci in range(ncases):
    ah, aw = int(rng.integers(3, 8)), int(rng.integers(3, 8))
    an = int(rng.integers(1, 3)) if min(ah, aw) >= 5 else 1
    if min(ah, aw) >= 7 and rng.random() < 0.3: an = 3
    Hs, Ws = int(rng.integers(56, 90)), int(rng.integers(56, 90))
    grey = rng.random() < 0.25 or os.environ.get("SWEEP_GREY") is not None
    Cc = 1 if grey else 3
    sigma = float(rng.choice([10.0, 25.0, 50.0]))
    major = "row" if rng.random() < 0.6 else "col"
    k = int(rng.choice([8, 8, 12, 16])) if an == 1 else 8
    nSim, nDisp, p = int(rng.integers(4, 7)), int(rng.integers(1, 3)), int(rng.integers(3, 6))
    N1, N2 = int(rng.choice([1, 2, 4, 8])), int(rng.choice([2, 4, 8, 16]))
    t2a = str(rng.choice(["id", "dct", "bior"] if k != 12 else ["id", "dct"]))
    t2b = str(rng.choice(["dct", "bior"] if k != 12 else ["dct"]))
    t4 = str(rng.choice(["sadct", "dct", "id"]))
    t5 = str(rng.choice(["haar", "hw", "dct"]))
    p1, p2 = (N1, nSim, nDisp, k, p, t2a, t4, t5), (N2, nSim, nDisp, 8 if k == 16 else k, p, t2b, t4, t5)
    lf = Hh.textured_lf(ah, aw, Hs, Ws)
    if grey:
        lf = np.ascontiguousarray(lf[:, :1])
    if major == "col":
        lf = np.ascontiguousarray(lf.reshape(ah, aw, Cc, Hs, Ws).transpose(1, 0, 2, 3, 4)).reshape(ah * aw, Cc, Hs, Ws)
    mask = np.ones(ah * aw, np.uint32)
    for _ in range(int(rng.integers(0, 3))):
        mask[int(rng.integers(0, ah * aw))] = 0
    cen = (ah // 2) * aw + aw // 2 if major == "row" else (ah // 2) + (aw // 2) * ah
    mask[cen] = 1
    lanes = int(rng.choice([1, 3]))
    os.environ["LFBM5D_LANES"] = str(lanes)
    mo, mg = (O.ROWMAJOR, L.ROWMAJOR) if major == "row" else (O.COLMAJOR, L.COLMAJOR)
    if only >= 0 and ci != only:
        continue
    tag = f"case{ci} {ah}x{aw} an{an} {Hs}x{Ws} C{Cc} s{sigma:g} {major} lanes{lanes} p1={p1} p2={p2} holes={int((mask == 0).sum())}"
    try:
        clean, noisy = Hh.noisy_lf(lf, sigma)
        noisy[mask == 0] = 0
        t0 = time.time()
        n1, b_o, st1 = O.run_step1(O.make_params(sigma, 2.7, *p1), noisy.copy(), mask, mo, aw, ah, an, Ws, Hs, Cc)
        w1_o = O.last_windows()
        n2, _, d_o, st2 = O.run_step2(O.make_params(sigma, 2.7, *p2), n1.copy(), b_o.copy(), mask, mo, aw, ah, an, Ws, Hs, Cc)
        w2_o = O.last_windows()
        t_or = time.time() - t0
        d_noisy = torch.from_numpy(noisy).cuda()
        d_basic, d_den = torch.zeros_like(d_noisy), torch.zeros_like(d_noisy)
        ctx.reset_stats()
        ctx.step1(core.make_params(sigma, 2.7, *p1), d_noisy, mask, d_basic, mg, aw, ah, an, Ws, Hs, Cc)
        w1_g, s1 = ctx.last_windows(), ctx.stats()
        b_g = d_basic.cpu().numpy()
        ctx.reset_stats()
        ctx.step2(core.make_params(sigma, 2.7, *p2), d_noisy, mask, d_basic, d_den, mg, aw, ah, an, Ws, Hs, Cc)
        w2_g, s2 = ctx.last_windows(), ctx.stats()
        d_g = d_den.cpu().numpy()
        if os.environ.get("SWEEP_LANES_CHECK"):   # the same steps on the other lane count: bit-identical?
            os.environ["LFBM5D_LANES"] = "1" if lanes == 3 else "3"
            e_basic, e_den = torch.zeros_like(d_noisy), torch.zeros_like(d_noisy)
            e_noisy = torch.from_numpy(noisy).cuda()
            ctx.step1(core.make_params(sigma, 2.7, *p1), e_noisy, mask, e_basic, mg, aw, ah, an, Ws, Hs, Cc)
            ctx.step2(core.make_params(sigma, 2.7, *p2), e_noisy, mask, e_basic, e_den, mg, aw, ah, an, Ws, Hs, Cc)
            print("   other lane count: basic identical", bool(torch.equal(e_basic, d_basic)), "denoised identical", bool(torch.equal(e_den, d_den)),
                  "| max |basic gpu - oracle|", float(np.abs(b_g - b_o).max()), "max |den gpu - oracle|", float(np.abs(d_g - d_o).max()),
                  "mean", float(np.abs(d_g - d_o).mean()))
            eo_, eg_ = np.abs(d_o - clean)[mask != 0], np.abs(d_g - clean)[mask != 0]
            print("   |denoised - clean| > 15 grey levels: oracle %d px (max %.1f), gpu %d px (max %.1f); noisy max err %.1f; both wrong at the same px: %d"
                  % ((eo_ > 15).sum(), eo_.max(), (eg_ > 15).sum(), eg_.max(), np.abs(noisy - clean)[mask != 0].max(), ((eo_ > 15) & (eg_ > 15)).sum()))
            bo_ = np.abs(b_o - clean)[mask != 0]
            print("   basic: |basic - clean| > 15: oracle %d px; exact zeros in oracle basic: %d, tiny (<1e-10, != 0): %d"
                  % ((bo_ > 15).sum(), (b_o[mask != 0] == 0).sum(), ((np.abs(b_o[mask != 0]) < 1e-10) & (b_o[mask != 0] != 0)).sum()))
            for st in range(0):
                if mask[st]:
                    print("    SAI", st, "PSNR gpu %.4f oracle %.4f" % (O.psnr_lf(d_g[st:st+1], clean[st:st+1]), O.psnr_lf(d_o[st:st+1], clean[st:st+1])),
                          "max|d|", float(np.abs(d_g[st] - d_o[st]).max()))
        m = mask != 0
        okw = np.array_equal(w1_g, w1_o) and np.array_equal(w2_g, w2_o)
        okp = (s1.windows, s1.passes) == (st1.windows, st1.passes) and (s2.windows, s2.passes) == (st2.windows, st2.passes)
        fin = np.isfinite(b_g).all() and np.isfinite(d_g).all() and np.isfinite(b_o).all() and np.isfinite(d_o).all()
        db = O.psnr_lf(b_g[m], clean[m]) - O.psnr_lf(b_o[m], clean[m])
        dd = O.psnr_lf(d_g[m], clean[m]) - O.psnr_lf(d_o[m], clean[m])
        worst = max(worst, abs(db), abs(dd))
        flag = "" if (okw and okp and fin and abs(db) < 0.02 and abs(dd) < 0.02) else "   <<<<<< CHECK"
        print(f"{tag}: windows {len(w1_g)}/{len(w2_g)} same={okw} passes_same={okp} [gpu {s1.passes},{s2.passes} oracle {st1.passes},{st2.passes}] finite={fin} dPSNR basic {db:+.4f} den {dd:+.4f} (oracle {t_or:.1f}s){flag}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{tag}: EXCEPTION {e}   <<<<<< CHECK", flush=True)
print("worst |dPSNR|", worst)
