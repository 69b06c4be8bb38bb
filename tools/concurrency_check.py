#!/usr/bin/env python3
"""Is a core pass bit-reproducible while OTHER lfbm5d contexts keep the same GPU busy?  (It is under foreign load --
torch GEMMs on another stream; see DESIGN.md section 9 for what this shows for concurrent lfbm5d contexts.)
usage: python tools/concurrency_check.py [H] [step]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfbm5d_amd as L
from lfbm5d_amd import core, synth
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lf = synth.make_lf(3, 3, H, W).reshape(9, 3, H, W).astype(np.float32)
lf += 25.0 * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
pk = (16, 18, 6, 8, 4, "dct", "sadct", "haar") if step == 2 else (8, 18, 6, 16, 4, "id", "sadct", "haar")
P = core.make_params(25.0, 2.7, *pk)
nHW = 24
pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
Hb, Wb = pad.shape[2:]
noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
basic = 0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)
num = torch.zeros_like(noisy); den = torch.zeros_like(noisy)
mask = np.ones(9, np.uint32); proc = np.zeros(9, np.uint32)
ctx = L.Context(0)
def one():
    num.zero_(); den.zero_(); torch.cuda.synchronize()
    ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic if step == 2 else None, num, den, mask, proc, 4, 4)
    torch.cuda.synchronize()
    refs, idx, cnt, best, shape = ctx.last_bm(pk[0], 9, Wb * Hb)
    valid = np.arange(pk[0])[None, :] < cnt[:, None]
    return num.cpu().numpy().copy(), den.cpu().numpy().copy(), np.where(valid, idx, 0), cnt.copy(), best.copy(), shape.copy()
ref = one()
stop = False
def noise():
    c2 = L.Context(0)
    n2 = noisy.clone(); b2 = basic.clone(); nu2 = torch.zeros_like(noisy); de2 = torch.zeros_like(noisy)
    torch.cuda.synchronize()
    while not stop:
        c2.core_pass(step, P, 3, 3, Wb, Hb, 3, n2, b2 if step == 2 else None, nu2, de2, mask, proc, 4, 4)
for mode in ("quiet", "another LFBM5D context busy"):
    if mode != "quiet":
        th = threading.Thread(target=noise); th.start(); th2 = threading.Thread(target=noise); th2.start()
    bad = [0] * 6
    for _ in range(10):
        o = one()
        for q in range(6):
            if q in (4, 5):
                k = pk[3]
                a2 = np.delete(o[q].reshape(9, Hb, Wb)[:, 6:Hb-6-k+1, 6:Wb-6-k+1], 4, axis=0); b2 = np.delete(ref[q].reshape(9, Hb, Wb)[:, 6:Hb-6-k+1, 6:Wb-6-k+1], 4, axis=0)
                bad[q] += int(not np.array_equal(a2, b2))
            else:
                bad[q] += int(not np.array_equal(o[q], ref[q]))
    print(f"step {step} {mode}: passes (of 10) differing from the reference pass in num/den/self_idx/self_cnt/best/shape:", bad)
stop = True
