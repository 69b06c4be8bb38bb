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


def _window_cover(aw, ah, pst):
    ps, pt = int(pst) // aw, int(pst) % aw
    s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
    return [(s0 + s) * aw + (t0 + t) for s in range(3) for t in range(3)], (ps - s0) * 3 + (pt - t0)


def _graph_job(noisy, ah, aw, H, W, pks, team, me, group):
    """One rank's part of the graph form of a job on the light field `noisy` (A x C*H*W): the oracle's core pass stands in for the
    device kernels and gloo send / recv for RCCL's.  `team` = the global ranks that share the job (plan rank i = team[i]), `me` =
    this rank's index in it, `group` = their process group (None: the world).  The rank walks the nodes of lfbm5d_plan_job in
    ISSUE ORDER, runs the windows it owns on num / den (and a basic estimate) of its own, and handles the messages in their issue
    order -- sums of a shared SAI between consecutive touchers, basic estimates from the rank that finalised them.  Returns the
    last step's estimate and the basic estimate (both inverse colour-transformed, complete on every rank of the team) and counters."""
    import helpers as Hh
    from oracle import oracle as O
    from lfbm5d_amd import core
    lib = O.lib()
    n_steps, world, Cc, sigma = len(pks), len(team), 3, 25.0
    A = ah * aw
    nodes, msgs, info = core.plan_job(aw, ah, world, 1, an=(1,) * n_steps)
    order = np.argsort(nodes[:, 6])
    assert sorted(nodes[:, 6].tolist()) == list(range(len(nodes)))
    plan = core.plan_windows(aw, ah, 1)
    for sl in range(n_steps):
        assert [int(n[2]) for n in nodes if n[0] == sl] == plan.tolist()
    # what run_bm5d_1st_step does before the windows (bm5d.cpp:133): forward colour transform; the second step reads the
    # light field after the first step's closing inverse transform and its own forward transform (bm5d.cpp:713, :827)
    lf = [noisy.copy()]
    for st in range(A):
        lib.orc_color_transform(lf[0][st], O.OPP, W, H, Cc, 1)
    if n_steps == 2:
        lf.append(lf[0].copy())
        for st in range(A):
            lib.orc_color_transform(lf[1][st], O.OPP, W, H, Cc, 0)
            lib.orc_color_transform(lf[1][st], O.OPP, W, H, Cc, 1)
    num = [np.zeros_like(lf[0]) for _ in range(n_steps)]
    den = [np.zeros_like(lf[0]) for _ in range(n_steps)]
    basic = np.zeros_like(lf[0])
    last = [{}, {}]
    for i, n in enumerate(nodes):
        for st in _window_cover(aw, ah, n[2])[0]:
            last[int(n[0])][st] = i
    mi = 0
    n_sent = n_recv = 0
    for i in order:
        sl, pst, r = int(nodes[i][0]), int(nodes[i][2]), int(nodes[i][3])
        if r == me:
            pk = pks[sl]
            nHW = pk[1] + pk[2]
            Wb, Hb = W + 2 * nHW, H + 2 * nHW
            idx, cst_w = _window_cover(aw, ah, pst)
            wn, wb, wnum, wden = (np.zeros((9, Cc * Wb * Hb), np.float32) for _ in range(4))
            for j, st in enumerate(idx):
                lib.orc_symetrize(lf[sl][st], wn[j], W, H, Cc, nHW)
                if sl == 1:
                    lib.orc_symetrize(basic[st], wb[j], W, H, Cc, nHW)
                lib.orc_symetrize(num[sl][st], wnum[j], W, H, Cc, nHW)
                lib.orc_symetrize(den[sl][st], wden[j], W, H, Cc, nHW)
            Hh.oracle_pass(sl + 1, sigma, pk, wn, wb if sl == 1 else None, Wb, Hb, Cc, num=wnum, den=wden, cst=cst_w, pst=cst_w)
            for j, st in enumerate(idx):
                lib.orc_unsymetrize(num[sl][st], wnum[j], W, H, Cc, nHW)
                lib.orc_unsymetrize(den[sl][st], wden[j], W, H, Cc, nHW)
            if n_steps == 2 and sl == 0:      # SAIs whose first-step sums are final: estimate, inverse + forward colour transform
                for st in idx:
                    if last[0][st] == i:
                        basic[st] = np.where(den[0][st] != 0, num[0][st] / np.where(den[0][st] != 0, den[0][st], 1), lf[0][st])
                        lib.orc_color_transform(basic[st], O.OPP, W, H, Cc, 0)
                        lib.orc_color_transform(basic[st], O.OPP, W, H, Cc, 1)
        while mi < len(msgs) and msgs[mi][1] == i:            # the messages this window's result feeds
            kind, _, to_node, to_rank, st, _ = (int(v) for v in msgs[mi])
            mi += 1
            if kind == 0:
                assert int(nodes[to_node][3]) == to_rank != r and int(nodes[to_node][0]) == sl
            bufs = [num[sl][st], den[sl][st]] if kind == 0 else [basic[st]]
            if r == me:
                for bf in bufs:
                    dist.send(torch.from_numpy(bf), team[to_rank], group=group)
                n_sent += 1
            elif to_rank == me:
                for bf in bufs:
                    dist.recv(torch.from_numpy(bf), team[r], group=group)
                n_recv += 1
    assert mi == len(msgs)
    # every SAI's final sums live on the rank of the last window that touched it; basic estimates where they were finalised
    ls = n_steps - 1
    est = np.zeros_like(lf[0])
    for st in range(A):
        owner = int(nodes[last[ls][st]][3])
        if owner == me:
            sub = lf[0][st] if ls == 0 else basic[st]
            est[st] = np.where(den[ls][st] != 0, num[ls][st] / np.where(den[ls][st] != 0, den[ls][st], 1), sub)
        dist.broadcast(torch.from_numpy(est[st]), team[owner], group=group)
        lib.orc_color_transform(est[st], O.OPP, W, H, Cc, 0)
        if n_steps == 2:
            dist.broadcast(torch.from_numpy(basic[st]), team[int(nodes[last[0][st]][3])], group=group)
            lib.orc_color_transform(basic[st], O.OPP, W, H, Cc, 0)
    return est, basic, dict(plan=plan, n_win=int(sum(1 for n in nodes if n[3] == me)), n_sent=n_sent, n_recv=n_recv, n_msgs=len(msgs))


def _graph_worker(rank, world, port, q, n_steps):
    """One rank of the graph form of a job (one step, or both steps as lfbm5d_denoise_* runs them) over the whole world: see _graph_job."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as Hh
    from oracle import oracle as O
    # two rows of windows for two ranks, three for three
    ah, aw = (5, 7) if world == 2 else (7, 11)
    H, W, Cc, sigma = 40, 40, 3, 25.0
    pks = [(4, 5, 2, 8, 4, "id", "sadct", "haar"), (8, 4, 2, 8, 3, "dct", "sadct", "haar")][:n_steps]
    clean, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, H, W), sigma)
    mask = np.ones(ah * aw, np.uint32)
    est, basic, info = _graph_job(noisy, ah, aw, H, W, pks, list(range(world)), rank, None)
    # the single-process oracle (data-driven windows, one rank): the two reference calls one after the other
    n_o, b_o, st1 = O.run_step1(O.make_params(sigma, 2.7, *pks[0]), noisy.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, Cc)
    same_plan = bool(np.array_equal(O.last_windows(), info["plan"]))
    if n_steps == 1:
        ok = bool(np.array_equal(est, b_o)); err = float(np.abs(est - b_o).max())
    else:
        _, b2_o, d_o, _ = O.run_step2(O.make_params(sigma, 2.7, *pks[1]), n_o, b_o, mask, O.ROWMAJOR, aw, ah, 1, W, H, Cc)
        ok = bool(np.array_equal(est, d_o) and np.array_equal(basic, b2_o)); err = float(max(np.abs(est - d_o).max(), np.abs(basic - b2_o).max()))
    q.put((rank, ok, err, info["n_win"], info["n_sent"], info["n_recv"], info["n_msgs"], same_plan))
    dist.barrier()
    dist.destroy_process_group()


def _banded_worker(rank, world, port, q):
    """Spatial bands x window graph (option spatial_bands, lfbm5d_steps.hip run_denoise_banded) on four gloo ranks: two teams of two.
    Team b runs the two-step job on band b of every SAI (its rows + a halo) as a graph job of its own on a process group of its own
    (_graph_job); then ONE all-gather over the world stitches the light fields, member t of a team contributing the t-th share of
    its band's rows -- the exchange run_denoise_banded does with ncclAllGather."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as Hh
    from oracle import oracle as O
    S, T = 2, world // 2
    groups = [dist.new_group(list(range(b * T, (b + 1) * T))) for b in range(S)]   # (every rank creates every group)
    ah, aw, H, W, Cc, sigma = 5, 7, 96, 40, 3, 25.0
    pks = [(4, 5, 2, 8, 4, "id", "sadct", "haar"), (8, 4, 2, 8, 3, "dct", "sadct", "haar")]
    # halo: at least nSim + nDisp + k of the wider step (the library's default), rounded up so that a band's first row keeps the
    # phase of BOTH steps' reference grids (p = 4 and 3): the band then processes the reference patches the whole field does
    halo = 24
    assert halo >= max(pk[1] + pk[2] + pk[3] for pk in pks) and (H // S - halo) % 12 == 0
    A = ah * aw
    clean, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, H, W), sigma)
    mask = np.ones(A, np.uint32)
    band, t = rank // T, rank % T
    y0, y1 = band * H // S, (band + 1) * H // S
    c0, c1 = max(0, y0 - halo), min(H, y1 + halo)
    img = lambda a, h: a.reshape(A, Cc, h, W)
    crop = np.ascontiguousarray(img(noisy, H)[:, :, c0:c1, :]).reshape(A, -1)
    est_c, basic_c, info = _graph_job(crop, ah, aw, c1 - c0, W, pks, list(range(band * T, (band + 1) * T)), t, groups[band])
    # inside a team the graph is exact: the band's job equals the single-process oracle on the same crop, bit for bit
    n_o, b_o, _ = O.run_step1(O.make_params(sigma, 2.7, *pks[0]), crop.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, c1 - c0, Cc)
    _, b2_o, d_o, _ = O.run_step2(O.make_params(sigma, 2.7, *pks[1]), n_o, b_o, mask, O.ROWMAJOR, aw, ah, 1, W, c1 - c0, Cc)
    team_exact = bool(np.array_equal(est_c, d_o) and np.array_equal(basic_c, b2_o))
    # the all-gather: equal chunks of rows, rank order = (band, member)
    rows = (H + S - 1) // S
    chunk = (rows + T - 1) // T
    ya = min(y1, y0 + t * chunk)
    n = min(y1, ya + chunk) - ya
    send = np.zeros((2, A, Cc, chunk, W), np.float32)
    send[0, :, :, :n] = img(est_c, c1 - c0)[:, :, ya - c0:ya - c0 + n]
    send[1, :, :, :n] = img(basic_c, c1 - c0)[:, :, ya - c0:ya - c0 + n]
    parts = [torch.zeros(send.shape) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(send))
    den_all, bas_all = np.zeros((A, Cc, H, W), np.float32), np.zeros((A, Cc, H, W), np.float32)
    for rk in range(world):
        b_, t_ = rk // T, rk % T
        q0, q1 = b_ * H // S, (b_ + 1) * H // S
        a_ = min(q1, q0 + t_ * chunk)
        m = min(q1, a_ + chunk) - a_
        den_all[:, :, a_:a_ + m] = parts[rk].numpy()[0, :, :, :m]
        bas_all[:, :, a_:a_ + m] = parts[rk].numpy()[1, :, :, :m]
    # against the whole-field single-process result: not bit-identical (a band's tables start their recurrence at its first row),
    # PSNR within BASELINE.json's tolerance
    n_w, b_w, _ = O.run_step1(O.make_params(sigma, 2.7, *pks[0]), noisy.copy(), mask, O.ROWMAJOR, aw, ah, 1, W, H, Cc)
    _, b2_w, d_w, _ = O.run_step2(O.make_params(sigma, 2.7, *pks[1]), n_w, b_w, mask, O.ROWMAJOR, aw, ah, 1, W, H, Cc)
    ps = lambda x: O.psnr_lf(x.reshape(A, -1), clean)
    q.put((rank, team_exact, ps(den_all) - ps(d_w), ps(bas_all) - ps(b2_w), float(np.abs(den_all.reshape(A, -1) - d_w).mean()), info["n_win"], info["n_sent"]))
    dist.barrier()
    dist.destroy_process_group()


def test_spatial_bands_times_window_graph_on_four_ranks():
    """2 bands x 2-rank window graph on gloo world-4 (the round-5 review's item 3): every team's job is bit-identical to the
    single-process oracle on its crop, every rank ends with the same stitched light field, all four ranks own windows and the
    teams exchange messages.  Against the whole-field result the stitched field is NOT bit-identical, at any halo: a band's
    distance tables start their float recurrences at its first row, near-tie matches and threshold decisions re-roll, and later
    windows build on them (with halo 48 = 2 (2 nHW + k), beyond the reach of both steps' border effects, the 144 x 40 field still
    differs by 0.11 grey levels on average).  The PSNR moves by the noise of the sample: +-0.03 dB on these 35 x 96 x 40 pixels
    (bound here 0.05 dB, mean |difference| 0.5 grey levels); on the device at 512 x 512 the same comparison holds 0.01 dB with
    room (tests/test_gpu_denoise.py test_spatial_bands_stay_within_the_psnr_tolerance: < 1e-3 dB measured)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_banded_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res                                  # team jobs exact on their crops
    assert all(abs(r[2]) < 0.05 and abs(r[3]) < 0.05 and r[4] < 0.5 for r in res), res   # stitched PSNR, denoised and basic; mean |difference|
    assert len({(round(r[2], 9), round(r[3], 9), round(r[4], 9)) for r in res}) == 1   # every rank ends with the same light fields
    assert all(r[5] > 0 for r in res) and sum(r[6] for r in res) > 0    # all four ranks own windows; messages travel inside the teams


@pytest.mark.parametrize("world,n_steps", [(2, 1), (3, 1), (2, 2), (3, 2)])
def test_graph_form_with_messages_equals_the_single_rank_step(world, n_steps):
    """world_size-2 / -3 gloo run of the multi-GPU scheme (graph form) -- windows owned per rank, one message per SAI a window
    needs from another rank's window, estimates formed by the last toucher -- for one step and for the TWO-STEP job
    (lfbm5d_denoise_*: second-step windows start as soon as their SAIs' basic estimates are final, which travel from the rank
    that finalised them): bit-identical to the oracle's single-process run_step1 (+ run_step2), every planned message used
    exactly once."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q, n_steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and same_plan for _, ok, _, _, _, _, _, same_plan in res), res
    assert sum(1 for _, _, _, n_win, *_ in res if n_win > 0) >= 2          # several ranks own windows (a rank the plan does not need only relays nothing)
    assert sum(r[4] for r in res) == sum(r[5] for r in res) == res[0][6] > 0   # every message sent once, received once


def test_band_count_rule_and_scale_model():
    """core.auto_bands (what bench.py --bands auto sets): the graph alone up to the ranks it keeps busy, bands for the rest, never a
    band narrower than twice its halo, always a divisor of the rank count; and tools/scale_model.py's band model picks the same
    counts as its best form at 8 ranks (host only: lfbm5d_plan_job)."""
    from lfbm5d_amd import core
    assert [core.auto_bands(17, 17, 512, 40, w) for w in (1, 2, 4, 8)] == [1, 1, 1, 2]
    assert [core.auto_bands(15, 15, 434, 40, w) for w in (1, 2, 4, 8)] == [1, 1, 1, 2]
    assert [core.auto_bands(9, 9, 512, 40, w) for w in (1, 2, 4, 8)] == [1, 1, 2, 4]
    for a in (3, 5, 9, 13, 17):
        for H in (96, 256, 512):
            for w in (1, 2, 3, 4, 6, 8):
                s = core.auto_bands(a, a, H, 40, w)
                assert s >= 1 and w % s == 0 and (s == 1 or H // s >= 80), (a, H, w, s)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_model as M
    for (a, H, W) in ((17, 512, 512), (9, 512, 512)):
        tg, tf, *_ = M.simulate(a, a, 8, 2.62, 2.47, H, W, 50.0, 20.0)
        best = min((sum(M.simulate_bands(a, a, 8, S, 2.62, 2.47, H, W, 50.0, 20.0, 40, 24)[:2]), S) for S in (2, 4) if H // S >= 80)
        assert best[0] < tg + tf                                  # at eight ranks the banded form beats the graph alone ...
        assert best[1] == core.auto_bands(a, a, H, 40, 8)         # ... and the rule picks the model's band count


def test_graph_plan_properties():
    """Host-side properties of the graph form for 1..8 ranks, one step and both: every window has an owner, the simulated execution respects the dependencies (a window starts after every earlier window of its step it shares an
    SAI with; a second-step window after the last first-step window on each of its SAIs), the issue order is the start order,
    and messages connect exactly the consecutive touchers of an SAI that live on different ranks (+ one basic estimate per SAI
    and reading rank in two-step jobs)."""
    from lfbm5d_amd import core
    for (ah, aw) in ((17, 17), (9, 9), (15, 15), (5, 7), (3, 3)):
        plan = core.plan_windows(aw, ah, 1)
        cover = [set(_window_cover(aw, ah, pst)[0]) for pst in plan]
        NW = len(plan)
        for world in (1, 2, 3, 4, 8):
            for lanes in (1, 3):
                ranks, lane, start = core.plan_graph(aw, ah, world, lanes)
                assert len(ranks) == NW and ranks.max() < world and lane.max() < lanes
                for w in range(NW):
                    for p in range(w):
                        if cover[w] & cover[p]:
                            assert start[p] < start[w], (ah, aw, world, lanes, p, w)
                # no two windows at once on one lane
                assert len({(int(ranks[w]), int(lane[w]), int(start[w])) for w in range(NW)}) == NW
            msgs = core.plan_messages(aw, ah, world)
            ranks, _, start = core.plan_graph(aw, ah, world)
            expect = []
            for w in range(NW):
                for st in sorted(cover[w]):
                    nxt = next((n for n in range(w + 1, NW) if st in cover[n]), None)
                    if nxt is not None and ranks[nxt] != ranks[w]:
                        expect.append((w, nxt, st))
            assert sorted((int(a), int(b), int(c)) for a, b, c, _ in msgs) == sorted(expect)
            key = [(int(start[int(m[0])]), int(m[0])) for m in msgs]
            assert key == sorted(key)                                 # issued by producer window, in start (= issue) order
            if world == 1:
                assert len(msgs) == 0
            # the two-step job
            for lanes in (1, 3):
                nodes, jm, info = core.plan_job(aw, ah, world, lanes, an=(1, 1))
                assert len(nodes) == 2 * NW and info["centre_ok"]
                pos = nodes[:, 6].astype(int)
                assert sorted(pos.tolist()) == list(range(2 * NW))
                st_ = nodes[:, 5].astype(int)
                assert all(st_[a] <= st_[b] for a, b in zip(np.argsort(pos)[:-1], np.argsort(pos)[1:]))   # issue order = start order
                cost = [10 if n[0] == 0 else 9 for n in nodes]
                last1 = {}
                for w in range(NW):
                    for st in cover[w]:
                        last1[st] = w
                for i in range(2 * NW):
                    sl, w = int(nodes[i][0]), int(nodes[i][1])
                    assert i == sl * NW + w
                    for p in range(w):
                        if cover[w] & cover[p]:
                            assert st_[sl * NW + p] + cost[sl * NW + p] <= st_[i]
                    if sl == 1:
                        for st in cover[w]:
                            assert st_[last1[st]] + 10 <= st_[i]
                assert len({(int(n[3]), int(n[4]), int(n[5])) for n in nodes}) == 2 * NW
                if lanes == 1:
                    rk = nodes[:, 3].astype(int)
                    expect = []
                    for sl in range(2):
                        for w in range(NW):
                            for st in sorted(cover[w]):
                                nxt = next((n for n in range(w + 1, NW) if st in cover[n]), None)
                                if nxt is not None and rk[sl * NW + nxt] != rk[sl * NW + w]:
                                    expect.append((0, sl * NW + w, sl * NW + nxt, int(rk[sl * NW + nxt]), st))
                    for st, w in last1.items():
                        readers = sorted({int(rk[NW + n]) for n in range(NW) if st in cover[n]} - {int(rk[w])})
                        expect += [(1, w, 0xffffffff, r, st) for r in readers]
                    assert sorted(tuple(int(v) for v in m[:5]) for m in jm) == sorted(expect)
                    assert [int(pos[int(m[1])]) for m in jm] == sorted(int(pos[int(m[1])]) for m in jm)
    r17, _, t17 = core.plan_graph(17, 17, 8)
    assert t17.max() + 1 <= 24                                        # critical path of the 17x17 backward raster
    # the two-step job is what lets eight ranks work: all of them busy, and a simulated makespan below a quarter of the serial time
    for (a, bound, busy) in ((17, 4.5, 8), (15, 3.5, 7)):
        nodes, _, info = core.plan_job(a, a, 8, 1, an=(1, 1))
        assert len(set(nodes[:, 3].tolist())) >= busy
        assert sum(10 if n[0] == 0 else 9 for n in nodes) / info["makespan"] >= bound


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


def _simulate_exchange(aw, ah, world, lanes, single_channel=False, perturb=False, n_steps=1):
    """Replay of what run_graph enqueues on every rank (lfbm5d_graph.hip): every rank walks the nodes of the job in ISSUE ORDER;
    a window goes to its lane's stream (FIFO) and waits for the previous toucher of each of its SAIs -- an event of the same
    rank, or the arrival of that SAI's message -- and, in the second step of a two-step job, for each SAI's basic estimate
    (finalised behind a window of the same rank, or arrived as a message); after a window, the messages it feeds are enqueued on
    the channel's exchange stream of the sender (gated by the window's completion) and of the receiver, in the list's order on
    both.  A send / recv pair completes when both are at the head of their streams (rendezvous).  Returns the number of
    operations that never complete (0: the issue order cannot deadlock)."""
    from lfbm5d_amd import core
    plan = core.plan_windows(aw, ah, 1)
    NW = len(plan)
    nodes, msgs, _ = core.plan_job(aw, ah, world, lanes, an=(1,) * n_steps)
    msgs = [tuple(int(v) for v in m) for m in msgs]
    ranks, lane, pos = nodes[:, 3].astype(int), nodes[:, 4].astype(int), nodes[:, 6].astype(int)
    order = np.argsort(pos)
    cover = [sorted(_window_cover(aw, ah, pst)[0]) for pst in plan]
    NN = len(nodes)
    prev = []
    for i in range(NN):
        sl, w = divmod(i, NW)
        prev.append({st: next((sl * NW + p for p in range(w - 1, -1, -1) if st in cover[p]), None) for st in cover[w]})
    last1 = {}
    for w in range(NW):
        for st in cover[w]:
            last1[st] = w
    sum_msg = {(a, b, st): i for i, (k, a, b, r, st, ch) in enumerate(msgs) if k == 0}
    basic_msg = {(st, r): i for i, (k, a, b, r, st, ch) in enumerate(msgs) if k == 1}
    # streams: ("lane", rank, lane) -> windows; ("ch", rank, channel) -> (message index, role)
    streams = {}
    for i in order:
        streams.setdefault(("lane", int(ranks[i]), int(lane[i])), []).append(("win", int(i)))
    for i, (k, a, b, r, st, ch) in enumerate(msgs):
        c = 0 if single_channel else ch
        streams.setdefault(("ch", int(ranks[a]), c), []).append(("send", i))
        streams.setdefault(("ch", r, c), []).append(("recv", i))
    if perturb:   # negative control: one rank enqueues its receives of a channel in reverse order
        k = next(k for k, q in streams.items() if k[0] == "ch" and sum(1 for op in q if op[0] == "recv") > 1)
        rec = [op for op in streams[k] if op[0] == "recv"][::-1]
        streams[k] = [rec.pop(0) if op[0] == "recv" else op for op in streams[k]]
    head = {k: 0 for k in streams}
    win_done, msg_done = [False] * NN, [False] * len(msgs)

    def chan(i, kind):
        k, a, b, r, st, ch = msgs[i]
        return ("ch", int(ranks[a]) if kind == "send" else r, 0 if single_channel else ch)

    def at_head(kind, i):
        k = chan(i, kind)
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
                        if p is not None:
                            ok = ok and (win_done[p] if ranks[p] == ranks[i] else msg_done[sum_msg[(p, i, st)]])
                        if n_steps == 2 and i >= NW:
                            f = last1[st]
                            ok = ok and (win_done[f] if ranks[f] == ranks[i] else msg_done[basic_msg[(st, int(ranks[i]))]])
                    if not ok:
                        break
                    win_done[i] = True
                else:
                    a = msgs[i][1]
                    if not (win_done[a] and at_head("send", i) and at_head("recv", i)):
                        break
                    msg_done[i] = True
                    head[chan(i, "recv" if kind == "send" else "send")] += 1
                head[k] += 1
                progress = True
    return win_done.count(False) + msg_done.count(False)


def test_exchange_issue_order_cannot_deadlock():
    """The RCCL exchange of the window graph has not run between real ranks yet (no multi-GPU box): what can be shown on the
    host is that the order in which every rank enqueues windows, sends and receives admits a complete execution under FIFO
    streams and rendezvous send / recv -- for every rank count and lane count bench.py can be asked for, with two exchange
    channels and with the one-channel fallback (second communicator unavailable), for single steps and for the two-step job."""
    for (ah, aw) in ((17, 17), (9, 9), (15, 15), (5, 7), (7, 11)):
        for world in (2, 3, 4, 5, 8):
            for lanes in (1, 2, 3):
                for single in (False, True):
                    for n_steps in (1, 2):
                        stuck = _simulate_exchange(aw, ah, world, lanes, single, n_steps=n_steps)
                        assert stuck == 0, (ah, aw, world, lanes, single, n_steps, stuck)
    assert _simulate_exchange(17, 17, 4, 3, False, perturb=True) > 0      # the replay does see a broken order
    assert _simulate_exchange(17, 17, 8, 3, False, perturb=True, n_steps=2) > 0
