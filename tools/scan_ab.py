#!/usr/bin/env python3
"""Distance tables of the two generations of the block-matching scan, entry for entry: one core pass with round 2's
kernel (LFBM5D_SCAN_V1=1), the same pass with the ring-sharing kernel, raw disparity tables (un-skewed from either
layout), self-search scores and the selections compared bit for bit.
usage: python tools/scan_ab.py [H] [W] [step] [sigma] [p] [nDisp] [nSim]   -> one JSON line; exit code 1 on any difference"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfbm5d_amd as L
from lfbm5d_amd import core, synth

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else H
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 25.0
lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
lf += sigma * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
pstep = int(sys.argv[5]) if len(sys.argv) > 5 else 4          # reference-grid step p
nDisp_ = int(sys.argv[6]) if len(sys.argv) > 6 else 6
nSim_ = int(sys.argv[7]) if len(sys.argv) > 7 else 18
pk = (16, nSim_, nDisp_, 8, pstep, "dct", "sadct", "haar") if step == 2 else (8, nSim_, nDisp_, 16, pstep, "id", "sadct", "haar")
P = core.make_params(sigma, 2.7, *pk)
N, nSim, nDisp, k = pk[0], pk[1], pk[2], pk[3]
nHW = nSim + nDisp
pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
Hb, Wb = pad.shape[2:]
noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
basic = 0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)
num = torch.zeros_like(noisy); den = torch.zeros_like(noisy)
mask = np.ones(9, np.uint32); proc = np.zeros(9, np.uint32)
ctx = L.Context(0)
NT = 8 * (2 * nDisp + 1) ** 2
NR, NC = Hb - 2 * nDisp - (k - 1), Wb - 2 * nDisp - (k - 1)
ii, cc = np.meshgrid(np.arange(NR), np.arange(NC), indexing="ij")
# round 2's layout: [strip][row + lane][64]
SR1 = NR + 63
TS1 = ((NC + 63) // 64) * 64 * SR1
idx1 = ((cc // 64) * SR1 + ii + cc % 64) * 64 + cc % 64
# second generation: [strip][Q / 4][lane][Q % 4], Q = row + lane + 3, strips from column 1; column 0 behind the strips
SRq = ((NR - 1 + 63 + 15) // 16) * 4 + 1
NS2 = (NC - 1 + 63) // 64
TS2 = NS2 * SRq * 256 + SRq * 4
c1 = np.maximum(cc - 1, 0)
Q = ii + c1 % 64 + 3
idx2 = np.where(cc == 0, NS2 * SRq * 256 + ii, (((c1 // 64) * SRq + Q // 4) * 64 + c1 % 64) * 4 + Q % 4)


def one(mode):
    os.environ.pop("LFBM5D_SCAN_V1", None); os.environ.pop("LFBM5D_SCAN_FULL_TABLES", None)
    if mode == "v1": os.environ["LFBM5D_SCAN_V1"] = "1"
    if mode == "full": os.environ["LFBM5D_SCAN_FULL_TABLES"] = "1"
    num.zero_(); den.zero_(); torch.cuda.synchronize()
    ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic if step == 2 else None, num, den, mask, proc, 4, 4)
    torch.cuda.synchronize()
    ver = ctx.last_scan_version()
    refs, idx, cnt, best, shape = ctx.last_bm(N, 9, Wb * Hb)
    valid = np.arange(N)[None, :] < cnt[:, None]
    if ver == 3:   # combined form: the tables never reach memory
        tab = None
    else:
        ts, ix = (TS1, idx1) if ver == 1 else (TS2, idx2)
        tab = ctx.last_tables(NT * ts).reshape(NT, ts)[:, ix.ravel()].reshape(NT, NR, NC).copy()
    sc = ctx.last_scores(len(refs) * (2 * nSim + 1) ** 2).copy()
    global a_refs
    a_refs = refs.copy()
    return ver, np.where(valid, idx, 0), cnt.copy(), best.copy(), shape.copy(), tab, sc, num.cpu().numpy().copy(), den.cpu().numpy().copy()


a = one("v1")
b = one("full")
cmb = one("comb")
out = {"Hb": Hb, "Wb": Wb, "step": step, "versions": [a[0], b[0]], "combined_version": cmb[0]}
tneq = a[5].view(np.uint32) != b[5].view(np.uint32)
out["tables_differ"] = int(tneq.sum())
if tneq.any():
    w = np.argwhere(tneq)
    out["first_table_diffs(table,row,col)"] = w[:12].tolist()
    out["tables_with_diffs"] = int(len(np.unique(w[:, 0])))
    out["diff_rows_min_max"] = [int(w[:, 1].min()), int(w[:, 1].max())]
    out["diff_cols_min_max"] = [int(w[:, 2].min()), int(w[:, 2].max())]
    t, r, c = w[0]
    out["first"] = {"v1": float(a[5][t, r, c]), "v2": float(b[5][t, r, c])}
sneq = a[6].view(np.uint32) != b[6].view(np.uint32)
out["scores_differ"] = int(sneq.sum())
if sneq.any():
    Ns2 = (2 * nSim + 1) ** 2
    w = np.argwhere(sneq.reshape(-1, Ns2))
    out["first_score_diffs(ref,cand)"] = w[:12].tolist()
    out["first_score"] = {"v1": float(a[6].reshape(-1, Ns2)[w[0][0], w[0][1]]), "v2": float(b[6].reshape(-1, Ns2)[w[0][0], w[0][1]])}
    out["cands_with_diffs"] = np.unique(w[:, 1])[:40].tolist()
    out["refs_with_diffs"] = np.unique(w[:, 0])[:40].tolist()
    # where in round 2's scores does the new kernel's (wrong) value occur?
    A2 = a[6].reshape(-1, Ns2).view(np.uint32); B2 = b[6].reshape(-1, Ns2).view(np.uint32)
    found = []
    for (r, cnd) in w[:200:10]:
        hit = np.argwhere(A2 == B2[r, cnd])[:3].tolist()
        found.append({"at": [int(r), int(cnd)], "v1_has_it_at": hit})
    out["where"] = found
    # pattern of the differences for a few forward candidates (the value at the reference position itself)
    gR = len(np.unique(a_refs // Wb)); gC = len(a_refs) // gR
    pat = {}
    for (dj_, di_) in ((0, 0), (18, 0), (18, 1), (19, 3), (36, 18)):
        cnd = dj_ * (2 * nSim + 1) + di_
        m = sneq.reshape(-1, Ns2)[:, cnd].reshape(gR, gC)
        rel = np.abs(a[6].reshape(-1, Ns2)[:, cnd] - b[6].reshape(-1, Ns2)[:, cnd]).reshape(gR, gC)
        pat["dj%d_di%d" % (dj_, di_)] = {"rows_with_diffs": np.nonzero(m.any(1))[0][:40].tolist(), "cols_with_diffs": np.nonzero(m.any(0))[0][:40].tolist(),
                                          "n": int(m.sum()), "max_abs": float(rel.max()), "row1": [float(v) for v in rel[1][:8]], "v1_row1": [float(v) for v in a[6].reshape(-1, Ns2)[:, cnd].reshape(gR, gC)[1][:8]]}
    out["pattern"] = pat
    # host model of one self table (float32 recurrence of core:3327-3390) to locate where the new kernel's values come from
    if os.environ.get("SCAN_AB_MODEL"):
        img = pad[4, 0].astype(np.float32)
        f32 = np.float32
        model = {}
        for (dj_, di_) in ((0, 0), (19, 3)):
            dk_r, dk_c = di_, dj_ - nSim
            D = np.zeros((Hb, Wb), np.float32)
            D[nHW:Hb - nHW, nHW:Wb - nHW] = (img[nHW + dk_r:Hb - nHW + dk_r, nHW + dk_c:Wb - nHW + dk_c] - img[nHW:Hb - nHW, nHW:Wb - nHW]) ** 2
            S = np.zeros((Hb, Wb), np.float32)
            v = f32(0)
            for p_ in range(k):
                for q_ in range(k): v = f32(v + D[nHW + p_, nHW + q_])
            S[nHW, nHW] = v
            for j in range(nHW + 1, Wb - nHW):
                sm = S[nHW, j - 1]
                for p_ in range(k): sm = f32(sm + f32(D[nHW + p_, j - 1 + k] - D[nHW + p_, j - 1]))
                S[nHW, j] = sm
            for i in range(nHW + 1, Hb - nHW):
                sm = S[i - 1, nHW]
                for q_ in range(k): sm = f32(sm + f32(D[i - 1 + k, nHW + q_] - D[i - 1, nHW + q_]))
                S[i, nHW] = sm
                for j in range(nHW + 1, Wb - nHW):
                    t_ = f32(S[i, j - 1] + S[i - 1, j]); t_ = f32(t_ - S[i - 1, j - 1]); t_ = f32(t_ + D[i + k - 1, j + k - 1]); t_ = f32(t_ - D[i + k - 1, j - 1])
                    t_ = f32(t_ - D[i - 1, j + k - 1]); t_ = f32(t_ + D[i - 1, j - 1]); S[i, j] = t_
            cnd = dj_ * (2 * nSim + 1) + di_
            res = []
            for r in (gC + 1, gC + 2, 2 * gC + 1, 5 * gC + 7):
                y, x = int(a_refs[r]) // Wb, int(a_refs[r]) % Wb
                v1v, v2v = a[6].reshape(-1, Ns2)[r, cnd], b[6].reshape(-1, Ns2)[r, cnd]
                hit = np.argwhere(S.view(np.uint32) == np.float32(v2v).view(np.uint32))[:4].tolist()
                res.append({"ref_yx": [y, x], "model": float(S[y, x]), "v1": float(v1v), "v2": float(v2v), "v2_found_in_model_at": hit})
            model["dj%d_di%d" % (dj_, di_)] = res
        out["model"] = model
    thr2 = float(a[6].max())
    out["diffs_where_v2_is_prefill"] = int((sneq & (b[6] == thr2)).sum()); out["diffs_where_v1_is_prefill"] = int((sneq & (a[6] == thr2)).sum())
out["self_idx_differ"] = int((a[1] != b[1]).sum()); out["self_cnt_differ"] = int((a[2] != b[2]).sum())
ys, xs = slice(nDisp, Hb - nDisp - k + 1), slice(nDisp, Wb - nDisp - k + 1)
ba = np.delete(a[3].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0); bb = np.delete(b[3].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0)
sa = np.delete(a[4].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0); sb = np.delete(b[4].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0)
out["best_differ"] = int((ba != bb).sum()); out["shape_differ"] = int((sa != sb).sum())
out["num_equal"] = bool(np.array_equal(a[7], b[7])); out["den_equal"] = bool(np.array_equal(a[8], b[8]))
# the combined form (the default): selections and sums against round 2's kernel
bc = np.delete(cmb[3].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0); sc_ = np.delete(cmb[4].reshape(9, Hb, Wb)[:, ys, xs], 4, axis=0)
out["combined"] = {"best_differ": int((ba != bc).sum()), "shape_differ": int((sa != sc_).sum()), "self_idx_differ": int((a[1] != cmb[1]).sum()),
                   "self_cnt_differ": int((a[2] != cmb[2]).sum()), "scores_differ": int((a[6].view(np.uint32) != cmb[6].view(np.uint32)).sum()),
                   "num_equal": bool(np.array_equal(a[7], cmb[7])), "den_equal": bool(np.array_equal(a[8], cmb[8]))}
if out["combined"]["best_differ"]:
    w = np.argwhere(ba != bc)
    out["combined"]["first_best_diffs(slot,row,col)"] = w[:12].tolist()
    out["combined"]["rows_min_max"] = [int(w[:, 1].min()), int(w[:, 1].max())]; out["combined"]["cols_min_max"] = [int(w[:, 2].min()), int(w[:, 2].max())]
print(json.dumps(out))
cbad = any(out["combined"][k] for k in ("best_differ", "shape_differ", "self_idx_differ", "self_cnt_differ", "scores_differ")) or cmb[0] != 3
bad = out["tables_differ"] or out["scores_differ"] or out["self_idx_differ"] or out["best_differ"] or out["shape_differ"] or b[0] != 2 or cbad
sys.exit(1 if bad else 0)
