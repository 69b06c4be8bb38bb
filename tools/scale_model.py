#!/usr/bin/env python3
"""Predicted multi-GPU step time of the two-step job (lfbm5d_denoise_*) from MEASURED one-GPU window times and a stated link
model -- no multi-GPU box has been available to this build, so this is a model, not a measurement.

Inputs: the job's graph, owners, issue order and message list exactly as the library plans them (lfbm5d_plan_job, host only);
the time of one window pass of either step with the kernels alone on a GPU (sum of the per-kernel averages of a one-lane rocprofv3
run, e.g. profiles/r04_b_kernel_stats_lanes1.csv); a link model: every ordered pair of ranks has one xGMI link of `--gbs` GB/s
per direction and `--lat-us` microseconds per message (RCCL send / recv launch + rendezvous), messages between the same two ranks
queue on it.  A rank is one server that walks its windows in issue order (one lane: the model does not credit the 10-15 % two
lanes buy on one GPU); a window starts when the rank is free, its local predecessors are done and its messages have arrived; a
message leaves when its producer window is done.  At the end every rank receives the SAIs it does not own of both outputs
(basic, denoised) through its seven links.

Spatial bands (option spatial_bands, lfbm5d_steps.hip run_denoise_banded): S teams of T = ranks / S ranks, each team runs the graph
of the job on its band of rows (H / S + up to two halos); a window pass on a band costs the whole-field pass x (band rows + 2 nHW) /
(H + 2 nHW) (every kernel of the pass is linear in reference rows; measured, profiles/r06_i_band_pass_times.txt), its messages
shrink by the same factor, and the job ends with one all-gather of both outputs.  The table lists every S x T for a rank count
and marks the best.

  python tools/scale_model.py                      # 17x17, 15x15, 9x9 at 1/2/4/8 ranks, README parameters
  python tools/scale_model.py --t-ht 3.15 --t-wiener 2.72 --gbs 50 --lat-us 20
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lfbm5d_amd import core  # noqa: E402


def simulate(aw, ah, world, t_ht, t_wien, H, W, gbs, lat_us):
    nodes, msgs, info = core.plan_job(aw, ah, world, 1, an=(1, 1), cost=(int(round(t_ht * 10)), int(round(t_wien * 10))))
    n = len(nodes)
    cost = np.where(nodes[:, 0] == 0, t_ht, t_wien)
    order = np.argsort(nodes[:, 6])
    rank = nodes[:, 3].astype(int)
    sai_bytes = 3 * H * W * 4
    # messages per producer, in issue order; dependencies of every node
    out = {}
    for k, (kind, a, b, r, st, ch) in enumerate(msgs):
        out.setdefault(int(a), []).append((int(kind), int(b) if kind == 0 else -1, int(r), int(st)))
    NW = n // 2
    plan = [int(x) for x in nodes[:NW, 2]]
    cover = []
    for pst in plan:
        ps, pt = pst // aw, pst % aw
        s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
        cover.append([(s0 + s) * aw + (t0 + t) for s in range(3) for t in range(3)])
    last1 = {}
    for w in range(NW):
        for st in cover[w]:
            last1[st] = w
    deps = []
    for i in range(n):
        sl, w = divmod(i, NW)
        d = set()
        for st in cover[w]:
            p = next((sl * NW + q for q in range(w - 1, -1, -1) if st in cover[q]), None)
            if p is not None:
                d.add((p, st, 0))
            if sl == 1:
                d.add((last1[st], st, 1))
        deps.append(d)
    fin = np.zeros(n)
    rank_free = np.zeros(world)
    link_free = {}
    arrive = {}   # (kind, producer, sai, to_rank) -> time
    for i in order:
        r = rank[i]
        ready = rank_free[r]
        for (p, st, kind) in deps[i]:
            ready = max(ready, fin[p] if rank[p] == r else arrive[(kind, p, st, r)])
        fin[i] = ready + cost[i]
        rank_free[r] = fin[i]
        for (kind, to_node, to_rank, st) in out.get(int(i), []):
            nbytes = sai_bytes * (2 if kind == 0 else 1)
            t0 = max(fin[i], link_free.get((r, to_rank), 0.0))
            t1 = t0 + lat_us * 1e-3 + nbytes / (gbs * 1e6)          # ms
            link_free[(r, to_rank)] = t1
            arrive[(kind, int(i), st, to_rank)] = t1
    t_graph = float(fin.max())
    # final exchange: both outputs, every rank receives what it does not own through world - 1 links
    t_final = final_exchange(aw, ah, H, W, world, gbs, lat_us)
    busy = [float(cost[rank == r].sum()) for r in range(world)]
    return t_graph, t_final, busy, len(msgs), float(cost.sum())


def final_exchange(aw, ah, H, W, world, gbs, lat_us):
    if world <= 1:
        return 0.0
    recv = 2 * aw * ah * 3 * H * W * 4 * (world - 1) / world
    return recv / (min(7, world - 1) * gbs * 1e6) + 2 * lat_us * 1e-3


def simulate_bands(aw, ah, world, S, t_ht, t_wien, H, W, gbs, lat_us, halo, nhw):
    """S bands x (world / S)-rank graphs; the widest band (an inner one: two halos) sets the time."""
    T = world // S
    rows = max(min(H, (b + 1) * H // S + halo) - max(0, b * H // S - halo) for b in range(S))
    f = (rows + 2 * nhw) / (H + 2 * nhw)
    tg, _, busy, nm, tot = simulate(aw, ah, T, t_ht * f, t_wien * f, rows, W, gbs, lat_us)
    return tg, final_exchange(aw, ah, H, W, world, gbs, lat_us), rows, f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t-ht", type=float, default=3.15, help="ms per window pass of the HT step, kernels alone on one GPU")
    ap.add_argument("--t-wiener", type=float, default=2.72, help="ms per window pass of the Wiener step")
    ap.add_argument("--gbs", type=float, default=50.0, help="xGMI GB/s per direction between two ranks (peak 76.8)")
    ap.add_argument("--lat-us", type=float, default=20.0, help="per-message overhead")
    ap.add_argument("--halo", type=int, default=40, help="rows of halo of a spatial band (library default: nSim + nDisp + k of the wider step = 40 with the README parameters)")
    ap.add_argument("--nhw", type=int, default=24, help="nSim + nDisp: rows of mirrored border a pass adds on either side")
    a = ap.parse_args()
    print(f"window pass {a.t_ht} / {a.t_wiener} ms (HT / Wiener), link {a.gbs} GB/s per direction + {a.lat_us} us per message")
    for (ah, aw, H, W) in ((17, 17, 512, 512), (15, 15, 434, 625), (9, 9, 512, 512)):
        base = None
        for world in (1, 2, 4, 8):
            tg, tf, busy, nm, tot = simulate(aw, ah, world, a.t_ht, a.t_wiener, H, W, a.gbs, a.lat_us)
            t = tg + tf
            base = base or t
            tg0, tf0 = tg, tf
            print(f"{ah}x{aw}x{H}x{W}  ranks {world}: windows {tg:7.1f} ms + final exchange {tf:4.1f} ms = {t:7.1f} ms  speed-up {base / t:4.2f}  "
                  f"busy ranks {sum(b > 0 for b in busy)}  messages {nm}  rank utilisation {min(busy) / tg:.2f}..{max(busy) / tg:.2f}")
            rows = []
            for S in (2, 4, 8):
                if S > world or H // S < 2 * a.halo:
                    continue
                tg, tf, r, f = simulate_bands(aw, ah, world, S, a.t_ht, a.t_wiener, H, W, a.gbs, a.lat_us, a.halo, a.nhw)
                rows.append((tg + tf, S, world // S, r, f, tg, tf))
            best = min(rows)[1] if rows else 0
            for (t, S, T, r, f, tg, tf) in rows:
                print(f"    bands {S} x graph {T}: band of {r} rows (pass x {f:.2f}): windows {tg:7.1f} ms + all-gather {tf:4.1f} ms = {t:7.1f} ms  "
                      f"speed-up {base / t:4.2f}{'   <- best, and better than the graph alone' if S == best and t < tg0 + tf0 else '   <- best banded' if S == best else ''}")


if __name__ == "__main__":
    main()
