"""world_size-2 gloo tests (CPU) of the multi-GPU schemes, with the oracle standing in for the device kernels:
 * whole steps, graph form: windows owned per rank + one message per SAI a window needs from another rank's window
   (lfbm5d_plan_graph / lfbm5d_plan_messages) -- bit-identical to the single-process step;
 * single core passes: each rank runs its shard of reference-patch rows, rank > 0 starts from zeroed aggregation
   buffers, an all-reduce(sum) of num/den restores base + all contributions -- what lfbm5d_pass_device does with RCCL."""
import os
import socket

import pytest
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as Hh
    from lfbm5d_amd import core
    lf = Hh.source_lf(crop=56)
    clean, noisy = Hh.noisy_lf(lf, 25.0)
    pk = (4, 5, 2, 8, 4, "dct", "sadct", "haar")
    win, Wb, Hb = Hh.padded_window(noisy, 56, 56, 3, 7)
    rng = np.random.default_rng(3)
    base_num = (rng.uniform(0, 5, size=win.shape)).astype(np.float32)   # buffers already hold earlier windows
    base_den = (rng.uniform(0.5, 1, size=win.shape)).astype(np.float32)
    full_num, full_den, st = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3, num=base_num.copy(), den=base_den.copy())
    n_rows = int(round(np.sqrt(st.groups)))
    b, e = core.shard_rows(n_rows, rank, world)
    # matching runs on the full base on every rank; only the accumulation starts from zero on ranks > 0
    num = base_num.copy() if rank == 0 else np.zeros_like(base_num)
    den = base_den.copy() if rank == 0 else np.zeros_like(base_den)
    if rank == 0:
        num, den, _ = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3, num=num, den=den, rows=(b, e))
    else:
        # BM must see base: emulate by running on base then subtracting it (additive outputs)
        n2, d2, _ = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3, num=base_num.copy(), den=base_den.copy(), rows=(b, e))
        num, den = n2 - base_num, d2 - base_den
    tn, td = torch.from_numpy(num), torch.from_numpy(den)
    dist.all_reduce(tn)
    dist.all_reduce(td)
    ok = np.allclose(tn.numpy(), full_num, rtol=1e-4, atol=1e-3) and np.allclose(td.numpy(), full_den, rtol=1e-4, atol=1e-4)
    q.put((rank, bool(ok), int(st.groups)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_pass_all_reduce_equals_full_pass():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _graph_worker(rank, world, port, q):
    """One rank of the graph form of a step, with the oracle's core pass standing in for the device kernels and gloo
    send / recv for RCCL's: the rank runs the windows lfbm5d_plan_graph gives it, in plan order, on num / den of its
    own, and handles the messages of lfbm5d_plan_messages in their issue order."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import helpers as Hh
    from oracle import oracle as O
    from lfbm5d_amd import core
    lib = O.lib()
    # two rows of windows for two ranks, three for three (a chain = a row of windows is the unit dealt to ranks)
    ah, aw = (5, 7) if world == 2 else (7, 11)
    H, W, Cc, sigma = 40, 40, 3, 25.0
    pk = (4, 5, 2, 8, 4, "id", "sadct", "haar")
    nHW = pk[1] + pk[2]
    A = ah * aw
    clean, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, H, W), sigma)
    mask = np.ones(A, np.uint32)
    plan = core.plan_windows(aw, ah, 1)                       # host-only entry points of the C-ABI library
    ranks, _, _ = core.plan_graph(aw, ah, world)
    msgs = core.plan_messages(aw, ah, world)
    # what run_bm5d_1st_step does before the windows (bm5d.cpp:133): forward colour transform
    lf = noisy.copy()
    for st in range(A):
        lib.orc_color_transform(lf[st], O.OPP, W, H, Cc, 1)
    num, den = np.zeros_like(lf), np.zeros_like(lf)
    Wb, Hb = W + 2 * nHW, H + 2 * nHW
    mi = 0
    n_sent = n_recv = 0
    for w, pst in enumerate(plan):
        if ranks[w] == rank:
            ps, pt = int(pst) // aw, int(pst) % aw
            cs_w, mins, maxs, ct_w, mint, maxt = (C.c_int() for _ in range(6))
            lib.orc_search_window(ps, ah, 1, C.byref(cs_w), C.byref(mins), C.byref(maxs))
            lib.orc_search_window(pt, aw, 1, C.byref(ct_w), C.byref(mint), C.byref(maxt))
            idx = [(mins.value + s) * aw + (mint.value + t) for s in range(3) for t in range(3)]
            wn, wnum, wden = (np.zeros((9, Cc * Wb * Hb), np.float32) for _ in range(3))
            for i, st in enumerate(idx):
                lib.orc_symetrize(lf[st], wn[i], W, H, Cc, nHW)
                lib.orc_symetrize(num[st], wnum[i], W, H, Cc, nHW)
                lib.orc_symetrize(den[st], wden[i], W, H, Cc, nHW)
            cst_w = cs_w.value * 3 + ct_w.value
            Hh.oracle_pass(1, sigma, pk, wn, None, Wb, Hb, Cc, num=wnum, den=wden, cst=cst_w, pst=cst_w)
            for i, st in enumerate(idx):
                lib.orc_unsymetrize(num[st], wnum[i], W, H, Cc, nHW)
                lib.orc_unsymetrize(den[st], wden[i], W, H, Cc, nHW)
        while mi < len(msgs) and msgs[mi][0] == w:            # the messages this window's result feeds
            _, to_w, st, _ = (int(v) for v in msgs[mi])
            mi += 1
            if ranks[w] == rank:
                dist.send(torch.from_numpy(num[st]), int(ranks[to_w])); dist.send(torch.from_numpy(den[st]), int(ranks[to_w]))
                n_sent += 1
            elif ranks[to_w] == rank:
                dist.recv(torch.from_numpy(num[st]), int(ranks[w])); dist.recv(torch.from_numpy(den[st]), int(ranks[w]))
                n_recv += 1
    # every SAI's final sums live on the rank of the last window that touched it
    last = {}
    for w, pst in enumerate(plan):
        ps, pt = int(pst) // aw, int(pst) % aw
        s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
        for s in range(3):
            for t in range(3):
                last[(s0 + s) * aw + (t0 + t)] = w
    est = np.zeros_like(lf)
    for st in range(A):
        if ranks[last[st]] == rank:
            est[st] = np.where(den[st] != 0, num[st] / np.where(den[st] != 0, den[st], 1), lf[st])
        t = torch.from_numpy(est[st])
        dist.broadcast(t, int(ranks[last[st]]))
        lib.orc_color_transform(est[st], O.OPP, W, H, Cc, 0)
    # the single-process oracle (data-driven windows, one rank)
    _, b_o, st1 = O.run_step1(O.make_params(sigma, 2.7, *pk), noisy.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, Cc)
    q.put((rank, bool(np.array_equal(est, b_o)), float(np.abs(est - b_o).max()), int(sum(1 for r in ranks if r == rank)), n_sent, n_recv,
           len(msgs), bool(np.array_equal(O.last_windows(), plan))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_graph_form_with_messages_equals_the_single_rank_step(world):
    """world_size-2 / -3 gloo run of the step-level multi-GPU scheme (graph form): windows owned per rank, one message per
    SAI a window needs from another rank's window, estimates formed by the last toucher -- bit-identical to the
    oracle's single-process step, and every planned message is used exactly once."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and same_plan for _, ok, _, _, _, _, _, same_plan in res), res
    assert all(n_win > 0 for _, _, _, n_win, *_ in res)                    # every rank owns windows
    assert sum(r[4] for r in res) == sum(r[5] for r in res) == res[0][6] > 0   # every message sent once, received once


def test_graph_plan_properties():
    """Host-side properties of the graph form for 1..8 ranks: every window has an owner, chains stay on one rank, the
    unit-time schedule respects the dependencies (a window starts after every earlier window it shares an SAI with),
    and messages connect exactly the consecutive touchers of an SAI that live on different ranks."""
    from lfbm5d_amd import core
    for (ah, aw) in ((17, 17), (9, 9), (15, 15), (5, 7), (3, 3)):
        plan = core.plan_windows(aw, ah, 1)
        cover = []
        for pst in plan:
            ps, pt = int(pst) // aw, int(pst) % aw
            s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
            cover.append({(s0 + s) * aw + (t0 + t) for s in range(3) for t in range(3)})
        for world in (1, 2, 3, 4, 8):
            for lanes in (1, 3):
                ranks, lane, start = core.plan_graph(aw, ah, world, lanes)
                assert len(ranks) == len(plan) and ranks.max() < world and lane.max() < lanes
                for w in range(len(plan)):
                    for p in range(w):
                        if cover[w] & cover[p]:
                            assert start[p] < start[w], (ah, aw, world, lanes, p, w)
                    if w and int(plan[w]) // aw == int(plan[w - 1]) // aw:
                        assert ranks[w] == ranks[w - 1]               # a chain (same row of SAIs) stays on one rank
                # no two windows at once on one lane
                assert len({(int(ranks[w]), int(lane[w]), int(start[w])) for w in range(len(plan))}) == len(plan)
            msgs = core.plan_messages(aw, ah, world)
            ranks, _, _ = core.plan_graph(aw, ah, world)
            expect = []
            for w in range(len(plan)):
                for st in sorted(cover[w]):
                    nxt = next((n for n in range(w + 1, len(plan)) if st in cover[n]), None)
                    if nxt is not None and ranks[nxt] != ranks[w]:
                        expect.append((w, nxt, st))
            assert sorted((int(a), int(b), int(c)) for a, b, c, _ in msgs) == sorted(expect)
            assert [int(m[0]) for m in msgs] == sorted(int(m[0]) for m in msgs)   # issued by producer window
            if world == 1:
                assert len(msgs) == 0
    r17, _, t17 = core.plan_graph(17, 17, 8)
    assert t17.max() + 1 <= 24                                        # critical path of the 17x17 backward raster


def test_plan_windows_matches_the_reference_rule():
    """Centre first, then always the last SAI not covered yet (bm5d.cpp:187-213 with all counts tied);
    the oracle's data-driven schedule runs the same number of windows."""
    from lfbm5d_amd import core
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    assert core.plan_windows(3, 3, 1).tolist() == [4]
    p55 = core.plan_windows(5, 5, 1)
    assert p55[0] == 12 and p55[1] == 24 and len(p55) == 5          # 2^2 + 1 windows
    assert len(core.plan_windows(17, 17, 1)) == 64 and len(core.plan_windows(9, 9, 1)) == 16
    col = core.plan_windows(5, 7, 1, core.COLMAJOR)
    assert col[0] == 3 + 2 * 7 and len(col) == len(core.plan_windows(5, 7, 1))
    mask = np.ones(25, np.uint32)
    mask[12] = 0                                                       # empty centre: starts from the last SAI
    assert core.plan_windows(5, 5, 1, mask=mask)[0] == 24
    for st in core.plan_windows(5, 5, 1, mask=mask):
        assert mask[st]


def _simulate_exchange(aw, ah, world, lanes, single_channel=False, perturb=False):
    """Replay of what run_step enqueues on every rank (lfbm5d_api.hip, graph form): every rank walks the windows in plan
    order; a window goes to its lane's stream (FIFO) and waits for the previous toucher of each of its SAIs -- an event of
    the same rank, or the arrival of that SAI's message; after a window, the messages it feeds are enqueued on the channel's
    exchange stream of the sender (gated by the window's completion) and of the receiver, in the list's order on both.
    A send / recv pair completes when both are at the head of their streams (rendezvous).  Returns the number of operations
    that never complete (0: the issue order cannot deadlock)."""
    from lfbm5d_amd import core
    plan = core.plan_windows(aw, ah, 1)
    ranks, lane, _ = core.plan_graph(aw, ah, world, lanes)
    msgs = [tuple(int(v) for v in m) for m in core.plan_messages(aw, ah, world)]
    cover = []
    for pst in plan:
        ps, pt = int(pst) // aw, int(pst) % aw
        s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
        cover.append(sorted((s0 + s) * aw + (t0 + t) for s in range(3) for t in range(3)))
    NW = len(plan)
    prev = [{st: next((p for p in range(w - 1, -1, -1) if st in cover[p]), None) for st in cover[w]} for w in range(NW)]
    msg_of = {(a, b, st): i for i, (a, b, st, _) in enumerate(msgs)}
    # streams: ("lane", rank, lane) -> windows; ("ch", rank, channel) -> (message index, role)
    streams = {}
    for w in range(NW):
        streams.setdefault(("lane", int(ranks[w]), int(lane[w])), []).append(("win", w))
    for i, (a, b, st, ch) in enumerate(msgs):
        c = 0 if single_channel else ch
        streams.setdefault(("ch", int(ranks[a]), c), []).append(("send", i))
        streams.setdefault(("ch", int(ranks[b]), c), []).append(("recv", i))
    if perturb:   # negative control: one rank enqueues its receives of a channel in reverse order
        k = next(k for k, q in streams.items() if k[0] == "ch" and sum(1 for op in q if op[0] == "recv") > 1)
        rec = [op for op in streams[k] if op[0] == "recv"][::-1]
        streams[k] = [rec.pop(0) if op[0] == "recv" else op for op in streams[k]]
    head = {k: 0 for k in streams}
    win_done, msg_done = [False] * NW, [False] * len(msgs)

    def at_head(kind, i):
        a, b, st, ch = msgs[i]
        k = ("ch", int(ranks[a] if kind == "send" else ranks[b]), 0 if single_channel else ch)
        return head[k] < len(streams[k]) and streams[k][head[k]] == (kind, i)
    progress = True
    while progress:
        progress = False
        for k, q in streams.items():
            while head[k] < len(q):
                kind, i = q[head[k]]
                if kind == "win":
                    ok = True
                    for st, p in prev[i].items():
                        if p is None:
                            continue
                        ok = ok and (win_done[p] if ranks[p] == ranks[i] else msg_done[msg_of[(p, i, st)]])
                    if not ok:
                        break
                    win_done[i] = True
                else:
                    a = msgs[i][0]
                    if not (win_done[a] and at_head("send", i) and at_head("recv", i)):
                        break
                    msg_done[i] = True
                    other = ("ch", int(ranks[msgs[i][1]] if kind == "send" else ranks[a]), 0 if single_channel else msgs[i][3])
                    head[other] += 1
                head[k] += 1
                progress = True
    return win_done.count(False) + msg_done.count(False)


def test_exchange_issue_order_cannot_deadlock():
    """The RCCL exchange of the window graph has not run between real ranks yet (no multi-GPU box): what can be shown on the
    host is that the order in which every rank enqueues windows, sends and receives admits a complete execution under FIFO
    streams and rendezvous send / recv -- for every rank count and lane count bench.py can be asked for, with two exchange
    channels and with the one-channel fallback (second communicator unavailable)."""
    for (ah, aw) in ((17, 17), (9, 9), (15, 15), (5, 7), (7, 11)):
        for world in (2, 3, 4, 5, 8):
            for lanes in (1, 2, 3):
                for single in (False, True):
                    stuck = _simulate_exchange(aw, ah, world, lanes, single)
                    assert stuck == 0, (ah, aw, world, lanes, single, stuck)
    assert _simulate_exchange(17, 17, 4, 3, False, perturb=True) > 0      # the replay does see a broken order
