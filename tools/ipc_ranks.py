#!/usr/bin/env python3
"""First-contact insurance for the multi-rank exchange on a ONE-GPU box (round 5): the window graph of a denoise between PROCESSES that
share the GPU, messages through IPC-mapped device buffers (lfbm5d_comm_init_ipc; RCCL refuses two ranks on one device) -- the same
issue order, event gating, channels and abort path as the RCCL form.  The parent runs the job on one rank, starts `world` fresh child
processes, and compares every rank's light fields with the single-rank result, bit for bit.

  python tools/ipc_ranks.py run <world> <case> [timeout_s]      case: 5x5 | 7x9 | 17x17x96
  python tools/ipc_ranks.py die <world> <case> [timeout_s]      rank world-1 leaves after the rendezvous: the others must return an error
                                                                within the watchdog (exit code 0 of THIS tool = they did), not hang
  python tools/ipc_ranks.py bands<S> <world> <case> [timeout_s] spatial bands (option spatial_bands = S): S teams of world / S processes, each team's job
                                                                on rendezvous names of its own, the stitch through IPC handles of the packed chunks; compared,
                                                                bit for bit, with the same banded job played by emulated ranks in the parent
prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401  (before the library: one HIP runtime per process, torch's)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "5x5": (5, 5, 64, 64, (4, 6, 2, 8, 4, "id", "sadct", "haar"), (8, 6, 2, 8, 4, "dct", "sadct", "haar")),
    "7x9": (7, 9, 64, 64, (4, 6, 2, 8, 4, "id", "sadct", "haar"), (8, 6, 2, 8, 4, "dct", "sadct", "haar")),
    "17x17x96": (17, 17, 96, 96, (8, 8, 3, 16, 4, "id", "sadct", "haar"), (16, 8, 3, 8, 4, "dct", "sadct", "haar")),
}


def light_field(case):
    from lfbm5d_amd import synth
    ah, aw, H, W, p1, p2 = CASES[case]
    lf = synth.make_lf(ah, aw, H, W).reshape(ah * aw, -1).astype(np.float32)
    return lf + 25.0 * np.random.default_rng(7).standard_normal(lf.shape).astype(np.float32)


def denoise(ctx, case, noisy):
    import torch
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    ah, aw, H, W, p1, p2 = CASES[case]
    P1, P2 = core.make_params(25.0, 2.7, *p1), core.make_params(25.0, 2.7, *p2)
    d_n = torch.from_numpy(noisy).cuda()
    d_b, d_o = torch.zeros_like(d_n), torch.zeros_like(d_n)
    ctx.reset_stats()
    ctx.denoise(P1, P2, d_n, np.ones(ah * aw, np.uint32), d_b, d_o, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
    torch.cuda.synchronize()
    st = ctx.stats()
    return d_n.cpu().numpy(), d_b.cpu().numpy(), d_o.cpu().numpy(), int(st.windows), int(st.messages)


def worker(rank, world, rdir, case, die, timeout_s, bands=1):
    import lfbm5d_amd as L
    ctx = L.Context(0)
    ctx.comm_init_ipc(rank, world, rdir, timeout_s)
    if bands > 1:
        ctx.set_option("spatial_bands", bands)
    if die and rank == world - 1:
        os._exit(0)          # gone after the rendezvous, before its first window
    noisy = light_field(case)
    t0 = time.time()
    try:
        n, b, o, windows, msgs = denoise(ctx, case, noisy)
    except L.LfBm5dError as e:
        print(json.dumps({"rank": rank, "error": str(e), "seconds": time.time() - t0}), flush=True)
        os._exit(7)
    # a second job on the same contexts: the gating words carry an epoch, buffers and handles are reused
    n2, b2, o2, _, _ = denoise(ctx, case, noisy)
    np.savez(os.path.join(rdir, f"out.{rank}.npz"), n=n, b=b, o=o, same_again=np.array([np.array_equal(n, n2) and np.array_equal(b, b2) and np.array_equal(o, o2)]),
             windows=np.array([windows]), msgs=np.array([msgs]))
    print(json.dumps({"rank": rank, "windows": windows, "messages": msgs, "seconds": time.time() - t0}), flush=True)


def main():
    mode = sys.argv[1]
    if mode == "worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6] == "1", float(sys.argv[7]), int(sys.argv[8]) if len(sys.argv) > 8 else 1)
    bands = int(mode[5:]) if mode.startswith("bands") else 1
    world, case = int(sys.argv[2]), sys.argv[3]
    timeout_s = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
    die = mode == "die"
    out = {"mode": mode, "world": world, "case": case}
    ref = None
    if not die:
        import lfbm5d_amd as L
        ctx = L.Context(0)
        if bands > 1:   # the reference of a banded job: the same teams played by emulated ranks (the bands are not bit-identical to one rank)
            ctx.set_option("emulate_world", world)
            ctx.set_option("spatial_bands", bands)
        ref = denoise(ctx, case, light_field(case))
        ctx.close()
        out["single_rank_windows"] = ref[3]
        out["spatial_bands"] = bands
    rdir = tempfile.mkdtemp(prefix="lfbm5d_ipc_")
    env = dict(os.environ)
    env.pop("LFBM5D_EMULATE_WORLD", None)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(world), rdir, case, "1" if die else "0", str(timeout_s), str(bands)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    codes, lines = [], []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout_s * 6 + 120)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            so += '\n{"error": "killed by the parent: no return within the limit"}'
        codes.append(p.returncode)
        lines.append([l for l in so.splitlines() if l.startswith("{")][-1:] or [se[-400:]])
    out["exit_codes"] = codes
    out["ranks"] = [json.loads(l[0]) if l and l[0].startswith("{") else {"stderr": l[0] if l else ""} for l in lines]
    out["seconds"] = time.time() - t0
    ok = True
    if die:
        # every surviving rank must have come back with an error, within the watchdog
        for r in range(world - 1):
            ok = ok and codes[r] == 7 and "error" in out["ranks"][r] and out["ranks"][r]["seconds"] < timeout_s * 4 + 30
        ok = ok and codes[world - 1] == 0
    else:
        total_windows = 0
        for r in range(world):
            f = os.path.join(rdir, f"out.{r}.npz")
            if codes[r] != 0 or not os.path.exists(f):
                ok = False
                continue
            z = np.load(f)
            same = bool(np.array_equal(z["n"], ref[0]) and np.array_equal(z["b"], ref[1]) and np.array_equal(z["o"], ref[2]))
            out["ranks"][r]["identical_to_single_rank"] = same
            out["ranks"][r]["second_job_identical"] = bool(z["same_again"][0])
            ok = ok and same and bool(z["same_again"][0])
            total_windows += int(z["windows"][0])
        out["windows_over_ranks"] = total_windows
        # (a banded job: every TEAM runs the whole schedule, and every member reports its team's windows)
        ok = ok and (total_windows == ref[3] if bands == 1 else total_windows > 0) and (world == 1 or case == "5x5" or bands == world or all(rk.get("messages", 0) > 0 for rk in out["ranks"]))
    out["ok"] = bool(ok)
    print(json.dumps(out))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
