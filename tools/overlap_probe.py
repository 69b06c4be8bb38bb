#!/usr/bin/env python3
"""Do core passes on several HIP streams overlap on one GPU, and do they leave each other's results alone?  Runs the headline window pass (HT or Wiener) from 1 and
from 2 contexts (own stream each, one host thread per context) and prints passes/s: the head-room a software
pipeline of block matching against transform+aggregation could reach.
usage: python tools/overlap_probe.py [step] [reps] [H]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfbm5d_amd as L  # noqa: E402
from lfbm5d_amd import core, synth  # noqa: E402


def worker(ctx, step, P, Wb, Hb, bufs, reps, barrier):
    noisy, basic, num, den = bufs
    mask = np.ones(9, np.uint32); proc = np.zeros(9, np.uint32)
    ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic, num, den, mask, proc, 4, 4)
    barrier.wait()
    for _ in range(reps):
        ctx.core_pass(step, P, 3, 3, Wb, Hb, 3, noisy, basic, num, den, mask, proc, 4, 4)


def main():
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    lf = synth.make_lf(3, 3, H, H).reshape(9, 3, H, H).astype(np.float32)
    lf += 25.0 * np.random.default_rng(1).standard_normal(lf.shape).astype(np.float32)
    pk = (8, 18, 6, 16, 4, "id", "sadct", "haar") if step == 1 else (16, 18, 6, 8, 4, "dct", "sadct", "haar")
    P = core.make_params(25.0, 2.7, *pk)
    nHW = 24
    pad = np.pad(lf, ((0, 0), (0, 0), (nHW, nHW), (nHW, nHW)), mode="symmetric")
    Hb, Wb = pad.shape[2:]
    for n_ctx in (1, 2, 3):
        ctxs = [L.Context(0) for _ in range(n_ctx)]
        bufs = []
        for _ in range(n_ctx):
            noisy = torch.from_numpy(np.ascontiguousarray(pad).reshape(9, -1)).cuda()
            basic = 0.5 * noisy + 0.5 * torch.roll(noisy, 1, 1)
            bufs.append((noisy, basic if step == 2 else None, torch.zeros_like(noisy), torch.zeros_like(noisy)))
        torch.cuda.synchronize()
        barrier = threading.Barrier(n_ctx + 1)
        th = [threading.Thread(target=worker, args=(ctxs[i], step, P, Wb, Hb, bufs[i], reps, barrier)) for i in range(n_ctx)]
        for t in th:
            t.start()
        barrier.wait()
        t0 = time.time()
        for t in th:
            t.join()
        dt = time.time() - t0
        torch.cuda.synchronize()
        sums = [(float(bf[2].double().sum()), float(bf[3].double().sum())) for bf in bufs]
        if n_ctx == 1:
            ref_sums = sums[0]
        same = all(sm == ref_sums for sm in sums)
        print(f"step {step}: {n_ctx} stream(s): {n_ctx * reps / dt:.1f} passes/s ({dt / reps * 1e3:.2f} ms per round of {n_ctx}); "
              f"results {'identical to the single-stream pass' if same else 'DIFFER: ' + str(sums) + ' vs ' + str(ref_sums)}")
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
