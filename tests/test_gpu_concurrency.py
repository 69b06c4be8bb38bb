"""Block matching must stay bit-exact while other lfbm5d contexts keep the same GPU busy.

Round 1 shipped a scan kernel whose 16-byte table stores could lose their first data register to the
next VALU instruction (gfx950 store-data hazard that hipcc does not pad when the store's scalar offset
is an SGPR; lfbm5d_bm.hip, store stage): alone on the GPU every pass was bit-reproducible, with two more
contexts running 8x8-patch passes a third of the passes returned wrong disparity arg-mins.  This test
is that scenario: three noise contexts, 50 Wiener (k = 8) and 20 HT (k = 16) passes on a 3x3x128^2
window, tables compared with the CPU oracle every time (core:3479-3611, :3301-3461).
"""
import threading

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_tables(est, Wb, Hb, pk, tau, refs):
    N, nSim, nDisp, k = pk[0], pk[1], pk[2], pk[3]
    o_idx = np.zeros((len(refs), max(N, 1)), np.uint32)
    o_cnt = np.zeros(len(refs), np.uint32)
    O.lib().orc_bm_self(np.ascontiguousarray(est[4]), Wb, Hb, k, N, nSim + nDisp, nSim, tau, refs, len(refs),
                        o_idx.reshape(-1), o_cnt)
    best, shape = {}, {}
    for st in range(9):
        if st == 4:
            continue
        ob, osh = np.zeros(Wb * Hb, np.uint32), np.zeros(Wb * Hb, np.uint8)
        O.lib().orc_bm_stereo(np.ascontiguousarray(est[4]), np.ascontiguousarray(est[st]), Wb, Hb, k, nDisp, tau, ob, osh)
        best[st], shape[st] = ob.reshape(Hb, Wb), osh.reshape(Hb, Wb)
    return o_idx, o_cnt, best, shape


@pytest.mark.parametrize("step,pk,passes", [(2, Hh.README_WIEN, 50), (1, Hh.README_HT, 20)], ids=["wiener-k8", "ht-k16"])
def test_block_matching_identical_to_oracle_under_concurrent_contexts(step, pk, passes):
    import lfbm5d_amd as L
    from lfbm5d_amd import core
    sigma, crop = 25.0, 128
    lf = Hh.source_lf(crop=crop)
    _, noisy = Hh.noisy_lf(lf, sigma)
    Cc = lf.shape[1]
    N, nSim, nDisp, k = pk[0], pk[1], pk[2], pk[3]
    win, Wb, Hb = Hh.padded_window(noisy, crop, crop, Cc, nSim + nDisp)
    # pilot for the Wiener pass: a smoothed copy (block matching only sees channel 0 of it)
    basic = np.ascontiguousarray((0.5 * win + 0.5 * np.roll(win, 1, axis=1)).astype(np.float32)) if step == 2 else None
    est = (win if step == 1 else basic)[:, :Wb * Hb]
    P = core.make_params(sigma, 2.7, *pk)
    mask, proc = np.ones(9, np.uint32), np.zeros(9, np.uint32)
    d_win = torch.from_numpy(win).cuda()
    d_basic = torch.from_numpy(basic).cuda() if basic is not None else None
    d_num, d_den = torch.zeros_like(d_win), torch.zeros_like(d_win)
    ctx = L.Context(0)

    def one():
        d_num.zero_(); d_den.zero_(); torch.cuda.synchronize()
        ctx.core_pass(step, P, 3, 3, Wb, Hb, Cc, d_win, d_basic, d_num, d_den, mask, proc, 4, 4)
        return ctx.last_bm(N, 9, Wb * Hb)

    refs = one()[0]
    o_idx, o_cnt, o_best, o_shape = _oracle_tables(est, Wb, Hb, pk, Hh.tau_match(sigma, Cc, step), refs)
    regr, regc = slice(nDisp, Hb - k - nDisp + 1), slice(nDisp, Wb - k - nDisp + 1)

    stop = threading.Event()
    errors = []

    def noise():
        try:
            c2 = L.Context(0)
            n2, nu2, de2 = d_win.clone(), torch.zeros_like(d_win), torch.zeros_like(d_win)
            b2 = d_basic.clone() if d_basic is not None else None
            torch.cuda.synchronize()
            while not stop.is_set():
                c2.core_pass(step, P, 3, 3, Wb, Hb, Cc, n2, b2, nu2, de2, mask, proc, 4, 4)
            c2.close()
        except Exception as e:   # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=noise) for _ in range(3)]
    for t in threads:
        t.start()
    try:
        for it in range(passes):
            _, idx, cnt, best, shape = one()
            assert np.array_equal(cnt, o_cnt), it
            valid = np.arange(max(N, 1))[None, :] < cnt[:, None]
            assert np.array_equal(np.where(valid, idx, 0), np.where(valid, o_idx, 0)), it
            for st in o_best:
                assert np.array_equal(best[st].reshape(Hb, Wb)[regr, regc], o_best[st][regr, regc]), (it, st)
                assert np.array_equal(shape[st].reshape(Hb, Wb)[regr, regc], o_shape[st][regr, regc]), (it, st)
    finally:
        stop.set()
        for t in threads:
            t.join()
        ctx.close()
    assert not errors, errors
