#!/usr/bin/env python3
"""The graph form of a step between REAL RCCL ranks on a one-GPU box: WORLD_SIZE processes that all use device 0 (RCCL
permitting -- several ranks per device is not a supported production layout, only a way to run the send / recv / broadcast
path of the exchange for real).  Every rank compares its result with a single-rank run of its own.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/two_ranks_one_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.distributed as dist
import lfbm5d_amd as L
from lfbm5d_amd import core
import helpers as Hh

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)     # rendezvous only; the exchange is the library's RCCL
ah, aw, Hs, Ws = 5, 7, 64, 64
clean, noisy = Hh.noisy_lf(Hh.textured_lf(ah, aw, Hs, Ws), 25.0)
mask = np.ones(ah * aw, np.uint32)
P1 = core.make_params(25.0, 2.7, 4, 6, 2, 8, 4, "id", "sadct", "haar")
P2 = core.make_params(25.0, 2.7, 8, 6, 2, 8, 4, "dct", "sadct", "haar")


def run(ctx):
    d_noisy = torch.from_numpy(noisy).cuda(); d_basic = torch.zeros_like(d_noisy); d_den = torch.zeros_like(d_noisy)
    ctx.reset_stats()
    ctx.step1(P1, d_noisy, mask, d_basic, L.ROWMAJOR, aw, ah, 1, Ws, Hs, 3)
    ctx.step2(P2, d_noisy, mask, d_basic, d_den, L.ROWMAJOR, aw, ah, 1, Ws, Hs, 3)
    return d_basic.cpu().numpy(), d_den.cpu().numpy(), ctx.stats()


solo = L.Context(0)
b0, d0, _ = run(solo)
solo.close()
ctx = L.Context(0)
ids = [L.Context.unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, 0)
try:
    ctx.comm_init(ids[0], rank, world)
    b, d, s = run(ctx)
    ok = bool(np.array_equal(b, b0) and np.array_equal(d, d0))
    print(f"rank {rank}/{world}: identical to the single-rank run: {ok}; windows {s.windows}, messages {s.messages}, comm {s.ms_comm:.2f} ms", flush=True)
except L.LfBm5dError as e:
    print(f"rank {rank}: RCCL refused ({e})", flush=True)
dist.barrier()
dist.destroy_process_group()
