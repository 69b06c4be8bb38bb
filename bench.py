#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LFBM5D core.

Metric (BASELINE.json): SAI-megapixels/s of the full HT + Wiener denoise, sigma = 25, on the
17x17x512x512 synthetic light field with the README "Stanford" parameters.
One "step" = run_bm5d_1st_step + run_bm5d_2nd_step over the whole light field, inputs already
resident in HBM, noise from the reference's MT19937 stream (seed 1).  The PSNR of the run is part of
the JSON line.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
  value      whole-job throughput of the timed region (default execution: the step's windows as a dependency
             graph on LFBM5D_LANES = 2 streams, bit-identical to the window-after-window order)
  roofline   the transform + aggregate kernel pair (k_group* + k_aggregate; the SURVEY 8d algorithmic bytes
             cover exactly that pair), durations from HIP events on the library's stream.  With lanes the
             kernels of different windows overlap on the GPU, so the pair is timed in one extra, untimed step
             with LFBM5D_LANES=1 (kernels alone on the GPU) right after the timed region; `--lanes 1` makes
             the timed region itself that measurement (the command the rocprof summaries under profiles/ use).
             `achieved`/`frac` follow the contract (algorithmic bytes / time / peak); `per_step` splits HT and
             Wiener and flags fractions above 1 (the byte model credits traffic the gather-form aggregation
             never makes); `traffic` / `frac_physical` are the PMC-measured HBM bytes of the same kernels
             (profiles/traffic_latest.json, collected with tools/collect_profiles.sh on this workload).
  seam       the same job through the reference's own seam, measured outside the timed region (the headline `value` stays the
             device-resident rate): `host_flat` = lfbm5d_denoise_host on pageable numpy buffers; `dropin_vectors` =
             run_bm5d_1st_step + run_bm5d_2nd_step of liblfbm5d_dropin.so on vector<vector<float>> light fields, i.e. the
             interval the reference times (main.cpp:189-201, :241-247), and run_bm5d (one job).  Since round 5 the SAIs are
             streamed through the window graph.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference: kind "port") on the GPU box's host
             cores: the first windows of each step of the SAME noisy light field (up to 3, bounded by a
             time limit), extrapolated by the window count; untiled parity mode.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s

WORKLOADS = {
    # README.md:60-61 "Stanford" parameters, aswSize 1, lambda 2.7, opp
    "lf17x17x512x512_sigma25": dict(ah=17, aw=17, H=512, W=512, sigma=25.0,
                                    p1=(8, 18, 6, 16, 4, "id", "sadct", "haar"),
                                    p2=(16, 18, 6, 8, 4, "dct", "sadct", "haar")),
    "lf9x9x512x512_sigma25": dict(ah=9, aw=9, H=512, W=512, sigma=25.0,
                                  p1=(8, 18, 6, 16, 4, "id", "sadct", "haar"),
                                  p2=(16, 18, 6, 8, 4, "dct", "sadct", "haar")),
    "lf3x3x256x256_sigma25": dict(ah=3, aw=3, H=256, W=256, sigma=25.0,
                                  p1=(8, 18, 6, 16, 4, "id", "sadct", "haar"),
                                  p2=(16, 18, 6, 8, 4, "dct", "sadct", "haar")),
    # the other BASELINE.json configurations (parity-test cases; selectable here to time them)
    "lf3x3x256x256_sigma25_dct": dict(ah=3, aw=3, H=256, W=256, sigma=25.0,            # configs[1]: dct/sadct/haar in both steps
                                      p1=(8, 18, 6, 16, 4, "dct", "sadct", "haar"),
                                      p2=(16, 18, 6, 8, 4, "dct", "sadct", "haar")),
    "lf17x17x512x512_sigma10_bior": dict(ah=17, aw=17, H=512, W=512, sigma=10.0,       # configs[3]: bior/sadct/haar
                                         p1=(8, 18, 6, 16, 4, "bior", "sadct", "haar"),
                                         p2=(16, 18, 6, 8, 4, "dct", "sadct", "haar")),
    "lf15x15x625x434_sigma50_n1": dict(ah=15, aw=15, H=434, W=625, sigma=50.0,         # configs[4]: EPFL-style, NHard = 1
                                       p1=(1, 18, 3, 16, 3, "bior", "sadct", "haar"),
                                       p2=(8, 18, 3, 8, 3, "dct", "sadct", "haar")),
}


def cpu_baseline(wl, noisy_rgb, basic_rgb, windows_per_step, total_mp, max_windows=3, time_limit=18.0):
    """Oracle (CPU restatement, OpenMP over reference patches, untiled) on the first windows of each step of the
    same noisy light field.  Checker code: imported and timed only here."""
    from oracle import oracle as O
    lib = O.lib()
    ah, aw, H, W = wl["ah"], wl["aw"], wl["H"], wl["W"]
    mask = np.ones(ah * aw, np.uint32)
    out = {"unit": "SAI-megapixels/s", "kind": "port", "cores": int(lib.orc_get_threads())}
    lib.orc_set_time_limit(float(time_limit))
    per_window, overhead, sampled = [], [], []
    results = {}   # the sampled windows' estimates, for the PSNR difference against the GPU on the same windows
    try:
        for step, pk in ((1, wl["p1"]), (2, wl["p2"])):
            P = O.make_params(wl["sigma"], 2.7, *pk)
            t0 = time.time()
            if step == 1:
                n_out, b_out, st = O.run_step1(P, noisy_rgb.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3, max_windows=max_windows)
                results["basic"] = (n_out, b_out)
            else:
                _, bs_out, d_out, st = O.run_step2(P, noisy_rgb.copy(), basic_rgb.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3,
                                                   max_windows=max_windows)
                results["denoised"] = (bs_out, d_out)      # SAIs no window touched keep the basic estimate
            wall = time.time() - t0
            n = max(1, int(st.windows))
            per_window.append(st.total_seconds / n)      # core passes (block matching + transforms + aggregation)
            overhead.append(max(0.0, wall - st.total_seconds))   # whole-LF colour transforms, padding, window choice
            sampled.append(n)
    finally:
        lib.orc_set_time_limit(0.0)
    est_total = sum(pw * windows_per_step for pw in per_window) + sum(overhead)
    out["value"] = total_mp / est_total
    # second leg: the reference's own parallel mode (OpenMP tiles with a discarded halo, bm5d.cpp:411-708; what the stock
    # CLI does with nbThreads = 0: nb_threads = largest power of two <= cores).  One window per step bounds the sample.
    try:
        tiles = 1
        while tiles * 2 <= out["cores"]:
            tiles *= 2
        if tiles > 1:
            lib.orc_set_tiles(tiles)
            tw = []
            for step, pk in ((1, wl["p1"]), (2, wl["p2"])):
                P = O.make_params(wl["sigma"], 2.7, *pk)
                if step == 1:
                    _, _, st = O.run_step1(P, noisy_rgb.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3, max_windows=1)
                else:
                    _, _, _, st = O.run_step2(P, noisy_rgb.copy(), basic_rgb.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, 3, max_windows=1)
                tw.append(st.total_seconds / max(1, int(st.windows)))
            t_total = sum(t * windows_per_step for t in tw) + sum(overhead)
            out["tiled"] = {"value": total_mp / t_total, "unit": "SAI-megapixels/s", "tiles": tiles, "cores": out["cores"],
                            "seconds_per_window_pass": {"ht": tw[0], "wiener": tw[1]},
                            "sample": (f"the reference's OpenMP tile mode ({tiles} tiles per SAI, halo of nSim+nDisp pixels computed and discarded; "
                                       f"about 0.5 dB below the untiled result): first window of each step, extrapolated to {windows_per_step} windows")}
    except Exception as e:
        out["tiled"] = {"value": None, "sample": f"failed: {e}"}
    finally:
        lib.orc_set_tiles(1)
    out["sample"] = (f"first {sampled[0]} (HT) + {sampled[1]} (Wiener) angular windows of the {ah}x{aw}x{H}x{W} steps on the same noisy "
                     f"light field: {per_window[0]:.2f} / {per_window[1]:.2f} s per window pass, + {sum(overhead):.1f} s of whole-LF work, "
                     f"extrapolated to {windows_per_step} windows per step (untiled parity mode, OpenMP over reference patches)")
    out["seconds_per_window_pass"] = {"ht": per_window[0], "wiener": per_window[1]}
    out["_results"] = results
    out["_sampled"] = sampled
    return out


def psnr_delta_vs_cpu(cb, clean, noisy0, basic_full, run_gpu):
    """BASELINE.json's "PSNR delta vs CPU ref": the GPU on exactly the windows the CPU leg ran (LFBM5D_MAX_WINDOWS, outside the
    timed region), both against the clean light field, over the SAIs those windows touched."""
    import torch
    res, sampled = cb.pop("_results"), cb.pop("_sampled")
    out = {}
    for key, n_win in (("basic", sampled[0]), ("denoised", sampled[1])):
        cpu_in, cpu_est = res[key]
        touched = np.nonzero((cpu_est != cpu_in).any(axis=1))[0]          # untouched SAIs keep the step's input
        os.environ["LFBM5D_MAX_WINDOWS"] = str(n_win)
        try:
            g_in, g_est = run_gpu(key, noisy0, basic_full)
        finally:
            os.environ.pop("LFBM5D_MAX_WINDOWS", None)
        g_in, g_est = g_in.cpu().numpy(), g_est.cpu().numpy()
        g_touched = np.nonzero((g_est != g_in).any(axis=1))[0]
        cl = clean.cpu().numpy()[touched]

        def psnr(x):
            mse = ((x.astype(np.float64) - cl) ** 2).mean(axis=1)
            return float((20 * np.log10(255.0 / np.sqrt(mse))).mean())
        pc, pg = psnr(cpu_est[touched]), psnr(g_est[touched])
        d = np.abs(cpu_est[touched].astype(np.float64) - g_est[touched])
        out[key] = {"windows": int(n_win), "touched_sais": int(len(touched)), "touched_sets_identical": bool(np.array_equal(touched, g_touched)),
                    "psnr_cpu_db": pc, "psnr_gpu_db": pg, "delta_db": pg - pc, "max_abs_diff": float(d.max()), "mean_abs_diff": float(d.mean()),
                    "pixels_compared": int(d.size), "pixels_off_by_more_than_1": int((d > 1.0).sum()),
                    "pixels_off_by_more_than_0p1": int((d > 0.1).sum())}
    return {"basic": out["basic"]["delta_db"], "denoised": out["denoised"]["delta_db"], "detail": out,
            "note": "GPU minus CPU (oracle, untiled), mean PSNR over the SAIs the sampled windows touched, same MT19937 noise; "
                    "the Wiener leg of both runs starts from the GPU's full basic estimate"}


def parity_vs_gpu(wl, noisy_rgb_9, basic_rgb_9, ctx):
    """--parity-check: the centre 3x3 window of the workload through the oracle AND through the C-ABI on the same padded
    window, one core pass per step: the checker's verdict in the bench line (coverage, estimate differences)."""
    import torch
    from oracle import oracle as O
    from lfbm5d_amd import core
    lib = O.lib()
    H, W = wl["H"], wl["W"]
    out = {}
    for step, pk, src in ((1, wl["p1"], None), (2, wl["p2"], basic_rgb_9)):
        P = O.make_params(wl["sigma"], 2.7, *pk)
        nHW = pk[1] + pk[2]
        Wb, Hb = W + 2 * nHW, H + 2 * nHW

        def pad(arr):
            o = np.zeros((9, 3 * Wb * Hb), np.float32)
            for st in range(9):
                im = np.ascontiguousarray(arr[st]).copy()
                lib.orc_color_transform(im, O.OPP, W, H, 3, 1)
                lib.orc_symetrize(im, o[st], W, H, 3, nHW)
            return o
        wn = pad(noisy_rgb_9)
        wb = pad(src) if src is not None else None
        num, den = np.zeros_like(wn), np.zeros_like(wn)
        mask, proc = np.ones(9, np.uint32), np.zeros(9, np.uint32)
        st = O.Stats()
        if lib.orc_pass(step, C.byref(P), 3, 3, Wb, Hb, 3, wn.reshape(-1), wb.ctypes.data if wb is not None else None,
                        num.reshape(-1), den.reshape(-1), mask, proc, 4, 4, 0, -1, C.byref(st)):
            raise RuntimeError("oracle pass failed")
        d_n = torch.from_numpy(wn).cuda()
        d_b = torch.from_numpy(wb).cuda() if wb is not None else None
        g_num, g_den = torch.zeros_like(d_n), torch.zeros_like(d_n)
        ctx.core_pass(step, core.make_params(wl["sigma"], 2.7, *pk), 3, 3, Wb, Hb, 3, d_n, d_b, g_num, g_den, mask, proc, 4, 4)
        gn, gd = g_num.cpu().numpy(), g_den.cpu().numpy()
        both = (den > 0) & (gd > 0)
        d = np.abs(num[both] / den[both] - gn[both] / gd[both])
        out["ht" if step == 1 else "wiener"] = {
            "coverage_identical": bool(np.array_equal(den > 0, gd > 0)),
            "mean_abs_estimate_diff": float(d.mean()), "p999_abs_estimate_diff": float(np.quantile(d, 0.999)),
            "max_abs_estimate_diff": float(d.max()),
            "psnr_between_estimates_db": float(10 * np.log10(255.0 ** 2 / max(float((d.astype(np.float64) ** 2).mean()), 1e-30)))}
    out["note"] = ("same padded window through the oracle and the C-ABI; the maximum belongs to the few hard-threshold decisions that fall "
                   "within float round-off of the threshold (float32 on the GPU, double accumulation in the oracle)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="lf17x17x512x512_sigma25", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seam", action="store_true", help="skip the host-seam measurements (`seam` in the line; +3 s)")
    ap.add_argument("--parity-check", action="store_true",
                    help="also run the centre window of each step through the oracle and the C-ABI and report the differences (+20 s)")
    ap.add_argument("--lanes", type=int, default=0, help="window lanes of the timed region (0: the library's default, 2)")
    ap.add_argument("--noise", default="mt19937", choices=["mt19937", "torch"],
                    help="mt19937: the reference's noise stream, seed 1 (default); torch: quick GPU noise for kernel iteration")
    ap.add_argument("--watchdog-s", type=int, default=600,
                    help="several GPUs: print an error line and exit if the run has not finished after this many seconds "
                         "(the RCCL exchange of the window graph has never run between real ranks: a hang should not be silent)")
    ap.add_argument("--job", default="auto", choices=["auto", "fused", "two-calls"],
                    help="fused: run_bm5d_1st_step + run_bm5d_2nd_step as ONE dependency graph of windows (lfbm5d_denoise_device; bit-identical "
                         "to the two calls, second-step windows start as soon as their SAIs' basic estimates are final); two-calls: "
                         "lfbm5d_step1_device then lfbm5d_step2_device; auto: fused")
    ap.add_argument("--sharding", default="graph", choices=["graph", "rows", "blocks"],
                    help="multi-GPU step scheme: graph = windows as a dependency graph, chains of windows per rank, one message per SAI "
                         "between ranks (default; bit-identical to one GPU); rows = row-sharded passes (exact); blocks = round 1's "
                         "contiguous window blocks + one all-reduce (scales, but not the reference's result)")
    ap.add_argument("--bands", default="auto",
                    help="several GPUs, fused job: spatial bands S (S teams of N / S ranks, each runs the window graph on a horizontal band of every "
                         "SAI + halo, one all-gather stitches the result; PSNR within 1e-3 dB of one GPU, not bit-identical).  auto: "
                         "lfbm5d_amd.core.auto_bands (1 up to the ranks the graph can keep busy); 1: the graph alone (bit-identical)")
    args = ap.parse_args()

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # several ranks: a rank's lanes (3 streams) and the two exchange streams should not share hardware queues -- a send
        # that waits for its peer must never sit in front of a compute stream on the same queue (ROCm maps streams onto
        # GPU_MAX_HW_QUEUES queues, 4 by default).  Must be set before the HIP runtime starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import lfbm5d_amd as L
    from lfbm5d_amd import core, synth

    if args.sharding in ("rows", "blocks"):
        os.environ["LFBM5D_STEP_SHARDING"] = args.sharding
    else:
        os.environ.pop("LFBM5D_STEP_SHARDING", None)
    if args.lanes > 0:
        os.environ["LFBM5D_LANES"] = str(args.lanes)
    lanes_timed = int(os.environ.get("LFBM5D_LANES", "2"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import threading
        import torch.distributed as dist

        def _give_up():
            if rank == 0:
                print(json.dumps({"metric": "SAI-megapixels/sec (HT+Wiener, sigma=25)", "value": None, "unit": "SAI-megapixels/s",
                                  "n_gpus": world, "error": f"no result after {args.watchdog_s} s with --sharding {args.sharding}: "
                                  "the multi-GPU exchange did not complete"}), flush=True)
            os._exit(3)
        wd = threading.Timer(args.watchdog_s, _give_up)
        wd.daemon = True
        wd.start()
        dist.init_process_group("nccl", rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    ah, aw, H, W, sigma = wl["ah"], wl["aw"], wl["H"], wl["W"], wl["sigma"]
    asize = ah * aw
    ctx = L.Context(local)
    if world > 1:
        idt = torch.zeros(core.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(L.Context.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
        if args.sharding == "graph":
            # pre-flight of the exchange between the real peers (all-reduce, split communicator, grouped send / recv on both
            # channels at once, broadcast).  An error on any rank moves every rank to the all-reduce scheme ("rows": exact,
            # slower) rather than losing the line; a hang ends in the watchdog above.
            try:
                ctx.comm_selftest(1 << 20)
                bad = 0
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: exchange pre-flight failed: {e}", file=sys.stderr, flush=True)
                bad = 1
            flag = torch.tensor([bad], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                args.sharding = "rows"
                os.environ["LFBM5D_STEP_SHARDING"] = "rows"

    # synthetic input, identical on every rank (the read-only light field is replicated)
    clean_u8 = synth.make_lf(ah, aw, H, W)
    clean_h = clean_u8.reshape(asize, -1).astype(np.float32)
    del clean_u8
    clean = torch.from_numpy(clean_h).cuda()
    noisy_h = None
    if args.noise == "mt19937":
        noisy_h = synth.add_noise_mt19937(clean_h, sigma, seed=1)     # utilities.cpp:176-183, one stream, st order
        noisy0 = torch.from_numpy(noisy_h).cuda()
    else:
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        noisy0 = clean + sigma * torch.randn(clean.shape, generator=g, device="cuda")
    del clean_h
    noisy = torch.empty_like(noisy0)
    basic = torch.zeros_like(noisy0)
    den = torch.zeros_like(noisy0)
    mask = np.ones(asize, np.uint32)
    P1 = core.make_params(sigma, 2.7, *wl["p1"])
    P2 = core.make_params(sigma, 2.7, *wl["p2"])
    FIELDS = ("windows", "passes", "groups", "stack_patches", "algorithmic_bytes", "ms_bm", "ms_group", "ms_aggregate",
              "ms_comm", "launches_group", "launches_aggregate", "lane_windows", "messages")

    fused = args.job in ("auto", "fused") and args.sharding == "graph"
    bands = 1
    if world > 1 and fused:
        halo = max(pk[1] + pk[2] + pk[3] for pk in (wl["p1"], wl["p2"]))      # the library's default halo: nSim + nDisp + k of the wider step
        bands = core.auto_bands(aw, ah, H, halo, world) if args.bands == "auto" else max(1, int(args.bands))
        ctx.set_option("spatial_bands", bands)

    def one_step(acc=None, two_calls=False):
        """HT + Wiener; acc: {"ht": {...}, "wiener": {...}} accumulates the library's counters per step kind (the fused job's
        counters all land under "ht": its kernels are not attributed to a step)."""
        noisy.copy_(noisy0)
        torch.cuda.synchronize()
        if fused and not two_calls:
            calls = (("ht", lambda: ctx.denoise(P1, P2, noisy, mask, basic, den, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)),)
        else:
            calls = (("ht", lambda: ctx.step1(P1, noisy, mask, basic, L.ROWMAJOR, aw, ah, 1, W, H, 3)),
                     ("wiener", lambda: ctx.step2(P2, noisy, mask, basic, den, L.ROWMAJOR, aw, ah, 1, W, H, 3)))
        for kind, call in calls:
            if acc is not None:
                ctx.reset_stats()
            call()
            if acc is not None:
                st = ctx.stats()
                for f in FIELDS:
                    acc[kind][f] = acc[kind].get(f, 0) + getattr(st, f)

    if bands > 1:
        # the banded form's team communicators (ncclCommSplit) and its all-gather have never met real ranks: one untimed job decides.  A
        # failure on any rank moves every rank to the graph alone (bit-identical, the form of rounds 3-5) rather than losing the line.
        try:
            one_step()
            bad = 0
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: banded job failed: {e}", file=sys.stderr, flush=True)
            bad = 1
        flag = torch.tensor([bad], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            bands = 1
            ctx.set_option("spatial_bands", 1)
            idt = torch.zeros(core.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")     # the failed job may have torn the communicators down
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(L.Context.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    timed = {"ht": {}, "wiener": {}}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(timed)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline measurement: the kernel pair alone on the GPU (one lane).  Outside the timed region.
    roof, roof_steps = timed, args.steps
    if (lanes_timed != 1 or fused) and world == 1:
        os.environ["LFBM5D_LANES"] = "1"
        roof, roof_steps = {"ht": {}, "wiener": {}}, 1
        one_step(roof, two_calls=True)
        torch.cuda.synchronize()
        os.environ["LFBM5D_LANES"] = str(lanes_timed)

    if rank == 0:
        total_mp = asize * H * W / 1e6
        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        value = total_mp * args.steps / elapsed

        def psnr_lf(x):   # quality: PSNR against the clean light field (mean over SAIs), this run
            mse = ((x - clean) ** 2).mean(dim=1)
            return float((20 * torch.log10(255.0 / torch.sqrt(mse))).mean().item())

        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if args.workload == tj.get("workload"):
                    traffic = tj
            except Exception:
                traffic = None

        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu_latest.json")
        if os.path.exists(vpath):
            try:
                vj = json.load(open(vpath))
                if args.workload == vj.get("workload"):
                    valu = vj
            except Exception:
                valu = None
        # SURVEY 8d's compulsory floor of a window pass: every window image read once, num / den read and written once
        comp = {}
        for kind, pk, S in (("ht", wl["p1"], 1), ("wiener", wl["p2"], 2)):
            nHW = pk[1] + pk[2]
            comp[kind] = (S + 4) * 4.0 * 9 * 3 * (W + 2 * nHW) * (H + 2 * nHW)
        SIMDS, CLK = 1024, 2.4e9      # 256 CUs x 4 SIMD-32; a wave64 VALU instruction occupies its SIMD for 2 cycles (MI355X_MICROARCH.md)
        HBM_ACHIEVABLE_GBS = 6300.0   # what a streaming kernel reaches on MI355X (MI355X_MICROARCH.md, HBM section): the floors' rate

        def valu_frac(classes, ms):
            """share of the chip's VALU issue slots the kernels of `classes` use during `ms` (their time alone on the GPU)"""
            if not valu or ms <= 0:
                return None
            k = valu.get("kernels", {})
            if any(c not in k for c in classes):
                return None
            insts = sum(k[c]["valu_insts_per_launch"] for c in classes)
            # valu_frac prices every instruction at 2 cycles; the packed-fp32 instructions these kernels are largely made of take 4
            # (measured, profiles/r06_a_valu_rate.txt: v_pk_fma / mul / add_f32 2.0-2.2 ns per wave-instruction and SIMD against 1.0-1.2
            # for the plain forms), as do DPP moves and v_mad_u32_u24: valu_frac_all_packed is the same count at 4 cycles -- the truth lies between
            return {"valu_insts": insts, "valu_frac": insts * 2.0 / (SIMDS * CLK * ms * 1e-3),
                    "valu_frac_all_packed": insts * 4.0 / (SIMDS * CLK * ms * 1e-3)}

        def pair(kind):
            a = roof[kind]
            n = max(1, int(a.get("launches_group", 0)))
            ms = (a.get("ms_group", 0.0) + a.get("ms_aggregate", 0.0)) / n
            by = a.get("algorithmic_bytes", 0.0) / n
            ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            d = {"algorithmic_bytes_per_launch": by, "avg_launch_ms": ms, "ms_group": a.get("ms_group", 0.0) / n,
                 "ms_aggregate": a.get("ms_aggregate", 0.0) / n, "launches": n, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
            if d["frac"] > 1.0:
                d["flag"] = ("above 1: the SURVEY 8d byte model charges 16 B of num/den read-modify-write per stacked pixel, "
                             "the gather-form aggregation moves 8 B (filt written once, read once) and the stack gathers hit L2")
            if traffic and kind in traffic.get("per_step", {}):
                tb = traffic["per_step"][kind]["hbm_bytes_per_launch"]
                d["traffic"] = tb
                d["frac_physical"] = tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None
                d["traffic_over_compulsory"] = tb / comp[kind]
            d["compulsory_bytes"] = comp[kind]
            # distance to the floor of THIS design (round 6; `frac` is saturated by the byte model): what each kernel class must move
            # through HBM once -- group: the window images in (the stack gathers re-read them from L2) + `filt` out; aggregation: `filt`
            # in + num / den in and out -- at the rate a streaming kernel reaches, next to the class's time alone on the GPU
            pk_, S_ = (wl["p1"], 1) if kind == "ht" else (wl["p2"], 2)
            nHW_, k_ = pk_[1] + pk_[2], pk_[3]
            plane_b = 4.0 * 9 * 3 * (W + 2 * nHW_) * (H + 2 * nHW_)
            n_pass = max(1, int(a.get("passes", 0)))
            filt_b = a.get("stack_patches", 0.0) / n_pass * 9 * 3 * k_ * k_ * 4.0
            design = {"group": S_ * plane_b + filt_b, "aggregate": filt_b + 4 * plane_b}
            d["filt_bytes"] = filt_b
            d["gather_bytes_from_l2"] = S_ * filt_b
            d["design_bytes"] = design["group"] + design["aggregate"]
            d["floor_ms"] = d["design_bytes"] / HBM_ACHIEVABLE_GBS / 1e6
            d["x_over_floor"] = ms / d["floor_ms"] if d["floor_ms"] > 0 else None
            d["floor_by_class"] = {cn: {"design_bytes": design[cn], "floor_ms": design[cn] / HBM_ACHIEVABLE_GBS / 1e6,
                                        "ms": d["ms_" + cn], "x_over_floor": (d["ms_" + cn] / (design[cn] / HBM_ACHIEVABLE_GBS / 1e6)) if design[cn] > 0 else None}
                                   for cn in ("group", "aggregate")}
            # what binds: per kernel class of this pass, the share of VALU issue slots it uses next to the share of the HBM peak it
            # moves (PMC bytes of the class / its time); neither near 1 = latency- / occupancy-bound
            n_p = max(1, int(a.get("passes", 0)))
            cls = {"block_matching": ((f"scan/{kind}", "select", "argmin"), a.get("ms_bm", 0.0) / n_p),
                   "group": ((f"group/{kind}",), d["ms_group"]), "aggregate": ((f"aggregate/{kind}",), d["ms_aggregate"])}
            bound = {}
            for cname, (classes, cms) in cls.items():
                e = {"ms": cms}
                v = valu_frac(classes, cms)
                if v:
                    e.update(v)
                if traffic and cms > 0:
                    tk = traffic.get("kernels", {})
                    by_c = 0.0
                    for c in classes:
                        kk = tk.get(c) or tk.get(c.split("/")[0] + "/both")
                        if kk:
                            by_c += kk["fetch_bytes_per_launch_x2"] + kk["write_bytes_per_launch"]   # FETCH_SIZE halves every shape (profiles/r06_b_fetch_calib.txt)
                    if by_c:
                        e["hbm_bytes"] = by_c
                        e["hbm_frac"] = by_c / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS
                if "valu_frac" in e or "hbm_frac" in e:
                    vf, hf = e.get("valu_frac", 0.0), e.get("hbm_frac", 0.0)
                    e["closest_bound"] = ("valu issue" if vf >= hf else "hbm") + f" at {max(vf, hf):.2f} -- " + (
                        "bound by it" if max(vf, hf) > 0.7 else "latency / occupancy: neither the VALUs nor HBM are saturated")
                bound[cname] = e
            d["bound_by_class"] = bound
            return d
        ph, pw = pair("ht"), pair("wiener")
        n_all = ph["launches"] + pw["launches"]
        alg_bytes = (ph["algorithmic_bytes_per_launch"] * ph["launches"] + pw["algorithmic_bytes_per_launch"] * pw["launches"]) / n_all
        pair_ms = (ph["avg_launch_ms"] * ph["launches"] + pw["avg_launch_ms"] * pw["launches"]) / n_all
        achieved = alg_bytes / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
        t_bytes = traffic.get("hbm_bytes_per_launch") if traffic else None
        tot = {f: timed["ht"].get(f, 0) + timed["wiener"].get(f, 0) for f in FIELDS}
        out = {
            "metric": "SAI-megapixels/sec (HT+Wiener, sigma=25)",
            "value": value, "unit": "SAI-megapixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "light_field": f"{ah}x{aw}x{H}x{W}x3", "sigma": sigma,
                       "params_ht": list(map(str, wl["p1"])), "params_wiener": list(map(str, wl["p2"])),
                       "asw": 1, "color_space": "opp", "noise": ("MT19937 seed 1 (utilities.cpp:176-183)" if args.noise == "mt19937" else "torch.randn seed 1"),
                       "window_lanes": lanes_timed,
                       "job": ("fused: both steps as one dependency graph of windows (lfbm5d_denoise_device), bit-identical to the two calls"
                               if fused else "two calls: lfbm5d_step1_device, lfbm5d_step2_device"),
                       "spatial_bands": bands,
                       "parallelism": ("single GPU" if world == 1 else
                                       f"{bands} spatial bands (rows of every SAI + halo, stitched by one RCCL all-gather; PSNR within 1e-3 dB of one GPU, not "
                                       f"bit-identical) x {world // bands} ranks per band on the dependency graph of both steps' angular windows (RCCL send/recv "
                                       "of num/den per shared SAI and of each SAI's basic estimate to the ranks that read it)" if bands > 1 else
                                       f"{world} ranks x chains of angular windows (dependency graph" + (" of both steps" if fused else "") + "), RCCL send/recv of num/den per shared SAI"
                                       + (" and of each SAI's basic estimate to the ranks that read it" if fused else "") +
                                       ", final broadcast of the estimates; bit-identical to one GPU" if args.sharding == "graph" else
                                       f"{world} x blocks of angular windows + 1 RCCL all-reduce of num/den per step (NOT the reference's result)"
                                       if args.sharding == "blocks" else
                                       f"{world} x reference-patch rows of every pass + RCCL all-reduce per pass")},
            "roofline": {"bound": "hbm", "kernel": "k_group* + k_aggregate (5-D transform + shrinkage + aggregation)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": t_bytes,
                         "frac_physical": (t_bytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (t_bytes and pair_ms > 0) else None,
                         "traffic_source": (traffic or {}).get("source"),
                         "compulsory_bytes": (comp["ht"] * ph["launches"] + comp["wiener"] * pw["launches"]) / n_all,
                         "valu_source": (valu or {}).get("source"),
                         "valu_frac": (lambda v: v["valu_frac"] if v else None)(valu_frac(("group/ht", "aggregate/ht", "group/wiener", "aggregate/wiener"), 2 * pair_ms)),
                         "valu_note": "VALU wave-instructions (rocprofv3 --pmc SQ_INSTS_VALU of this command, profiles/valu_latest.json) x 2 cycles / "
                                      "(1024 SIMDs x 2.4 GHz x live kernel time); per kernel class under per_step.*.bound_by_class",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": pair_ms, "launches": n_all,
                         # the contract's `frac` per step, and the distance to this design's own floor (what to read instead of a saturated frac)
                         "frac_ht": ph["frac"], "frac_wiener": pw["frac"],
                         "design_bytes": (ph["design_bytes"] * ph["launches"] + pw["design_bytes"] * pw["launches"]) / n_all,
                         "floor_ms": (ph["floor_ms"] * ph["launches"] + pw["floor_ms"] * pw["launches"]) / n_all,
                         "x_over_floor": pair_ms / ((ph["floor_ms"] * ph["launches"] + pw["floor_ms"] * pw["launches"]) / n_all) if (ph["floor_ms"] + pw["floor_ms"]) > 0 else None,
                         "x_over_floor_by_class": {"ht": {cn: v["x_over_floor"] for cn, v in ph["floor_by_class"].items()},
                                                   "wiener": {cn: v["x_over_floor"] for cn, v in pw["floor_by_class"].items()}},
                         "floor_note": "design_bytes = window images read once + filt written once (group), filt read once + num/den read and written "
                                       "once (aggregation); floor_ms = design_bytes / 6.3 TB/s (the rate MI355X_MICROARCH.md calls achievable); the stack "
                                       "gathers (gather_bytes_from_l2 under per_step) are served by L2 and are not in the floor",
                         "measured_with": ((f"the timed region ({lanes_timed} lane{'s' if lanes_timed != 1 else ''}"
                                            + ("; kernels of different windows overlap, the intervals are not kernel-alone times)" if lanes_timed != 1 else ")"))
                                           if roof is timed else
                                           f"{roof_steps} extra untimed step with LFBM5D_LANES=1, the two calls one after the other (kernels alone on the GPU); the timed region ran {lanes_timed} lanes" + (", fused job" if fused else "")),
                         "per_step": {"ht": ph, "wiener": pw}},
            # kernel classes alone on the GPU (the one-lane measurement step); with several lanes the HIP-event intervals of
            # different windows overlap and add up to more than the step, so those are reported under their own key
            "kernel_ms_per_step": {"block_matching": (roof["ht"].get("ms_bm", 0.0) + roof["wiener"].get("ms_bm", 0.0)) / roof_steps,
                                   "group": (roof["ht"].get("ms_group", 0.0) + roof["wiener"].get("ms_group", 0.0)) / roof_steps,
                                   "aggregate": (roof["ht"].get("ms_aggregate", 0.0) + roof["wiener"].get("ms_aggregate", 0.0)) / roof_steps,
                                   "comm": (roof["ht"].get("ms_comm", 0.0) + roof["wiener"].get("ms_comm", 0.0)) / roof_steps,
                                   "measured_with": (f"the timed region ({lanes_timed} lanes: overlapped intervals)" if (roof is timed and lanes_timed != 1)
                                                     else "the timed region (one lane)" if roof is timed else "the one-lane measurement step (kernels alone on the GPU)")},
            "kernel_ms_per_step_overlapped": {"block_matching": tot["ms_bm"] / args.steps, "group": tot["ms_group"] / args.steps,
                                              "aggregate": tot["ms_aggregate"] / args.steps, "comm": tot["ms_comm"] / args.steps,
                                              "note": "HIP-event intervals on each lane's stream during the timed region; intervals of "
                                                      "different windows overlap and add up to more than the step time"},
            "rccl_ranks_seen": ctx.comm_ranks() if world > 1 else 1,
            "passes_per_step": tot["passes"] / args.steps, "windows_per_step": tot["windows"] / args.steps,
            "lane_windows_per_step": tot["lane_windows"] / args.steps, "messages_per_step": tot["messages"] / args.steps,
            "psnr": {"noisy": psnr_lf(noisy0), "basic": psnr_lf(basic), "denoised": psnr_lf(den)},
        }
        if not args.no_seam and world == 1:
            # the job through the reference's own seam (host memory in and out), outside the timed region
            try:
                seam = {"note": "pageable host memory, SAIs streamed through the window graph (upload at first use, outputs behind the last "
                                "window on a SAI); value above = device-resident buffers; outputs compared with the timed region's"}
                n_src = noisy_h if noisy_h is not None else noisy0.cpu().numpy()
                den_ref = den.cpu().numpy()
                hb, hd = np.zeros_like(n_src), np.zeros_like(n_src)
                best = None
                for _ in range(2):
                    hn = n_src.copy()
                    t1 = time.perf_counter()
                    ctx.denoise(P1, P2, hn, mask, hb, hd, L.ROWMAJOR, aw, ah, 1, 1, W, H, 3)
                    dt = time.perf_counter() - t1
                    best = dt if best is None else min(best, dt)
                seam["host_flat"] = {"entry": "lfbm5d_denoise_host", "ms_per_step": best * 1e3, "value": total_mp / best,
                                     "vs_device": best * 1e3 / ms_per_step, "identical_to_device": bool(np.array_equal(hd, den_ref))}
                del hb, hd, hn
                for key, one_job in (("dropin_vectors", False), ("dropin_vectors_one_job", True)):
                    msv, _, _, dd = core.dropin_probe(n_src, mask, aw, ah, W, H, 3, sigma, 2.7, wl["p1"], wl["p2"], one_job=one_job, reps=2)
                    tt = float(msv[1].sum()) * 1e-3
                    seam[key] = {"entry": ("run_bm5d (both steps as one job)" if one_job else
                                           "run_bm5d_1st_step + run_bm5d_2nd_step (src/bm5d.h:11-62; the interval of main.cpp:189-201 + :241-247)"),
                                 "ms_per_step": tt * 1e3, "ms_calls": [float(msv[1, 0]), float(msv[1, 1])], "value": total_mp / tt,
                                 "vs_device": tt * 1e3 / ms_per_step, "identical_to_device": bool(np.array_equal(dd, den_ref))}
                    del dd
                out["seam"] = seam
            except Exception as e:  # noqa: BLE001
                out["seam"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the 1-GPU run only
            try:
                n_h = noisy_h if noisy_h is not None else noisy0.cpu().numpy()
                basic_full = basic.clone()
                out["cpu_baseline"] = cpu_baseline(wl, n_h, basic_full.cpu().numpy(), int(round(tot["windows"] / args.steps / 2)), total_mp)

                def run_gpu(key, n0, b_full):
                    n_in = n0.clone()
                    if key == "basic":
                        est = torch.zeros_like(n0)
                        ctx.step1(P1, n_in, mask, est, L.ROWMAJOR, aw, ah, 1, W, H, 3)
                    else:
                        b_in, est = b_full.clone(), torch.zeros_like(n0)
                        ctx.step2(P2, n_in, mask, b_in, est, L.ROWMAJOR, aw, ah, 1, W, H, 3)
                    torch.cuda.synchronize()
                    return (n_in if key == "basic" else b_in), est
                try:
                    out["cpu_baseline"]["psnr_delta_db"] = psnr_delta_vs_cpu(out["cpu_baseline"], clean, noisy0, basic_full, run_gpu)
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"].pop("_results", None); out["cpu_baseline"].pop("_sampled", None)
                    out["cpu_baseline"]["psnr_delta_db"] = {"basic": None, "denoised": None, "note": f"failed: {e}"}
                if args.parity_check:
                    cc = (ah // 2) * aw + aw // 2
                    idx = [cc + ds * aw + dt for ds in (-1, 0, 1) for dt in (-1, 0, 1)]
                    out["cpu_baseline"]["parity_vs_gpu"] = parity_vs_gpu(wl, n_h[idx], basic[idx].cpu().numpy(), ctx)
            except Exception as e:  # the baseline is a reported aside, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "SAI-megapixels/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
