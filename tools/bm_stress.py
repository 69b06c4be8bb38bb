#!/usr/bin/env python3
"""Block matching under concurrent lfbm5d contexts on one GPU: does a core pass return the same
tables as the same pass run alone?  Prints where disparity arg-mins differ (table row / column).
usage: python tools/bm_stress.py [H] [step] [passes] [noise_threads]"""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfbm5d_amd as L
from lfbm5d_amd import core, synth

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 30
nthreads = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
lf += 25.0 * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
pk = (16, 18, 6, 8, 4, "dct", "sadct", "haar") if step == 2 else (8, 18, 6, 16, 4, "id", "sadct", "haar")
P = core.make_params(25.0, 2.7, *pk)
nHW, nDisp, k = 24, 6, pk[3]
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
# round 2's kernel (LFBM5D_SCAN_V1=1): [strip][row + lane][64]
SR1 = NR + 63
TS1 = ((NC + 63) // 64) * 64 * SR1
IDX1 = ((cc // 64) * SR1 + ii + cc % 64) * 64 + cc % 64
# second generation: [strip][Q / 4][lane][Q % 4], Q = row + lane + 3, strips from column 1; column 0 behind the strips
SRq = ((NR - 1 + 63 + 15) // 16) * 4 + 1
NS2 = (NC - 1 + 63) // 64
TS2 = NS2 * SRq * 256 + SRq * 4
c1 = np.maximum(cc - 1, 0)
Q = ii + c1 % 64 + 3
IDX2 = np.where(cc == 0, NS2 * SRq * 256 + ii, (((c1 // 64) * SRq + Q // 4) * 64 + c1 % 64) * 4 + Q % 4)


def one():
    num.zero_(); den.zero_(); torch.cuda.synchronize()
    ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic if step == 2 else None, num, den, mask, proc, 4, 4)
    torch.cuda.synchronize()
    refs, idx, cnt, best, shape = ctx.last_bm(pk[0], 9, Wb * Hb)
    valid = np.arange(pk[0])[None, :] < cnt[:, None]
    ver = ctx.last_scan_version()
    if ver == 3:
        # combined form (the default): the disparity tables never reach memory -- `tables` holds (value, order) pairs per workgroup
        # and per-table edge arrays (include/lfbm5d.h, lfbm5d_last_scan_version).  The raw-table diff needs the full tables:
        # LFBM5D_SCAN_FULL_TABLES=1 (same kernel, table stores instead of the in-workgroup reduction); without it only the
        # arg-mins and selections are compared (pair entries whose row lies outside the table are never written).
        tab = np.zeros((1, 1, 1), np.float32)
    else:
        ts, ix = (TS1, IDX1) if ver == 1 else (TS2, IDX2)
        tab = ctx.last_tables(NT * ts).reshape(NT, ts)[:, ix.ravel()].reshape(NT, NR, NC).copy()   # un-skewed: [table][row][column]
    return np.where(valid, idx, 0), cnt.copy(), best.reshape(9, Hb, Wb).copy(), shape.reshape(9, Hb, Wb).copy(), tab


ref = one()
stop = False


def noise():
    c2 = L.Context(0)
    n2 = noisy.clone(); b2 = basic.clone(); nu2 = torch.zeros_like(noisy); de2 = torch.zeros_like(noisy)
    torch.cuda.synchronize()
    while not stop:
        c2.core_pass(step, P, 3, 3, Wb, Hb, 3, n2, b2 if step == 2 else None, nu2, de2, mask, proc, 4, 4)


out = {"lib": os.environ.get("LFBM5D_HIP_LIB", "default"), "H": H, "step": step}
for mode in ("quiet", "busy"):
    ths = []
    if mode == "busy":
        ths = [threading.Thread(target=noise) for _ in range(nthreads)]
        for t in ths: t.start()
    bad_self = bad_best = bad_tab = 0
    diffs = []; tdiffs = []
    for it in range(passes):
        o = one()
        bad_self += int(not (np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1])))
        ys, xs = slice(nDisp, Hb - nDisp - k + 1), slice(nDisp, Wb - nDisp - k + 1)
        a, b = o[2][:, ys, xs], ref[2][:, ys, xs]
        a = np.delete(a, 4, axis=0); b = np.delete(b, 4, axis=0)
        d = np.argwhere(a != b)
        if len(d):
            bad_best += 1
            for (sl, r, c) in d[:12]:
                diffs.append((it, int(sl), int(r), int(c), int(a[sl, r, c]) - int(b[sl, r, c])))
        # raw tables
        ta, tb = o[4], ref[4]
        neq = ta.view(np.uint32) != tb.view(np.uint32)
        td = np.argwhere(neq)
        if len(td):
            bad_tab += 1
            for (tbl, r, c) in td[:16]:
                ctxv = [float(v) for v in ta[tbl, r, max(c - 1, 0):c + 3]]
                refv = [float(v) for v in tb[tbl, r, max(c - 1, 0):c + 3]]
                # where else in the reference table does the wrong value occur?
                hit = np.argwhere(ref[4][tbl].view(np.uint32) == ta[tbl, r, c].view(np.uint32))[:3].tolist()
                tdiffs.append({"pass": it, "table": int(tbl), "row": int(r), "col": int(c), "got": ctxv, "exp": refv, "got_found_at(row,col)": hit, "n": int(len(td))})
    if mode == "busy":
        stop = True
        for t in ths: t.join()
    out[mode] = {"passes": passes, "bad_self": bad_self, "bad_best": bad_best, "bad_tab": bad_tab, "tdiffs": tdiffs[:40], "diffs(pass,slot,row,col,delta)": diffs[:60]}
print(json.dumps(out))
