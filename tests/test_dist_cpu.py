"""world_size-2 gloo test (CPU) of the multi-GPU scheme: each rank runs the core pass on its shard
of reference-patch rows (here with the oracle standing in for the device kernels), rank > 0
starts from zeroed aggregation buffers, an all-reduce(sum) of num/den restores base + all
contributions -- exactly what lfbm5d_pass_device does with RCCL."""
import os
import socket
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


def _window_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lfbm5d_amd import core
    aw, ah = 7, 5
    plan = core.plan_windows(aw, ah, 1)                       # host-only entry point of the C-ABI library
    b, e = len(plan) * rank // world, len(plan) * (rank + 1) // world   # run_step's block of this rank
    # stand-in for the per-rank aggregation buffers: every window adds 1 to the 3x3 SAIs it covers
    num = np.zeros((ah, aw), np.float32)
    for pst in plan[b:e]:
        ps, pt = int(pst) // aw, int(pst) % aw
        s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
        num[s0:s0 + 3, t0:t0 + 3] += 1
    tn = torch.from_numpy(num)
    dist.all_reduce(tn)                                        # the one all-reduce of a step
    full = np.zeros((ah, aw), np.float32)
    for pst in plan:
        ps, pt = int(pst) // aw, int(pst) % aw
        s0, t0 = min(max(ps - 1, 0), ah - 3), min(max(pt - 1, 0), aw - 3)
        full[s0:s0 + 3, t0:t0 + 3] += 1
    q.put((rank, bool(np.array_equal(tn.numpy(), full)), bool((full > 0).all()), len(plan), e - b))
    dist.barrier()
    dist.destroy_process_group()


def test_window_blocks_all_reduce_covers_every_sai():
    """world_size-2 gloo run of the step-level scheme: contiguous blocks of the planned window sequence per
    rank, one all-reduce of the per-rank sums."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_window_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and covered for _, ok, covered, _, _ in res)
    assert sum(n for *_, n in res) == res[0][3]


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
