"""CPU tests pinning the oracle end to end.

Pins: the mean/per-SAI PSNRs of the reference itself, recorded in SURVEY.md section 6 / BASELINE.md
(probe build of the unmodified reference run in the survey container, MT19937 seed 1 noise) on the
reference's only data fixture testing/sourceLF (committed as tests/golden/sourceLF_3x3_256_u8.npy).
The HT step reproduces them to 1e-6 dB; the Wiener step (2-D DCT through FFTW in the reference,
direct double-accumulated DCT here and in the probe's stand-in) to better than 1e-3 dB.
"""
import numpy as np
import pytest

from oracle import oracle as O
import helpers as Hh

# (sigma, HT params, Wiener params, reference noisy/basic/denoised mean PSNR) -- BASELINE.md section 2
PINS = {
    "readme": (25.0, Hh.README_HT, Hh.README_WIEN, (20.167152, 34.207336, 35.708221)),
    "config4": (10.0, Hh.C4_HT, Hh.README_WIEN, (28.125967, 36.227036, 40.121140)),
    "config5": (50.0, Hh.C5_HT, Hh.C5_WIEN, (14.146553, 30.035460, 31.998465)),
}
REF_BASIC_PER_SAI = [33.9496, 34.2953, 33.9713, 34.3323, 35.0939, 34.1320, 33.9865, 34.2174, 33.8877]
REF_DEN_PER_SAI = [35.3435, 35.8933, 35.3730, 35.9357, 36.6932, 35.7582, 35.3067, 35.7367, 35.3338]


@pytest.mark.parametrize("name", ["readme", "config4", "config5"])
def test_end_to_end_psnr_matches_reference_run(name):
    sigma, p1, p2, (ref_noisy, ref_basic, ref_den) = PINS[name]
    clean, noisy = Hh.noisy_lf(Hh.source_lf(), sigma)
    assert abs(O.psnr_lf(noisy, clean) - ref_noisy) < 2e-6
    mask = np.ones(9, np.uint32)
    noisy0 = noisy.copy()   # run_step1 mutates its LF_noisy argument in place, like the reference
    n1, basic, st1 = O.run_step1(O.make_params(sigma, 2.7, *p1), noisy, mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
    assert st1.windows == 1 and st1.passes == 1   # colour LF: one centre pass per window
    pb = O.psnr_lf(basic, clean)
    assert abs(pb - ref_basic) < 5e-6, pb
    n2, b2, den, st2 = O.run_step2(O.make_params(sigma, 2.7, *p2), n1, basic.copy(), mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
    pd = O.psnr_lf(den, clean)
    assert abs(pd - ref_den) < 1e-3, pd
    if name == "readme":
        np.testing.assert_allclose([O.psnr(basic[i], clean[i]) for i in range(9)], REF_BASIC_PER_SAI, atol=6e-5)
        np.testing.assert_allclose([O.psnr(den[i], clean[i]) for i in range(9)], REF_DEN_PER_SAI, atol=5e-4)
        assert st1.groups == 3721 and st2.groups == 3969          # SURVEY section 6 group statistics
        assert st1.sadct_groups + st2.sadct_groups == 4
        # step 1 leaves the caller's noisy LF drifted by the lossy OPP round trip
        d = np.abs(n1 - noisy0)
        assert 0.1 < d.max() < 1.0 and 0.02 < d.mean() < 0.3


def test_greyscale_runs_the_subset_path():
    """C == 1: the window is not done after the centre pass, further SAIs run the den-aware subset
    path (core:531-821); every SAI ends up covered."""
    lf = Hh.source_lf(crop=80)[:, :1]
    clean, noisy = Hh.noisy_lf(lf, 20.0)
    mask = np.ones(9, np.uint32)
    P = O.make_params(20.0, 2.7, 4, 6, 2, 8, 4, "dct", "sadct", "haar", cs="rgb")
    n1, basic, st = O.run_step1(P, noisy, mask, O.ROWMAJOR, 3, 3, 1, 80, 80, 1)
    assert st.windows == 1 and st.passes > 1
    assert O.psnr_lf(basic, clean) > O.psnr_lf(noisy, clean) + 3


def test_five_by_five_schedule():
    """5x5 colour LF: 2^2 + 1 windows, one centre pass each."""
    from lfbm5d_amd import synth
    lf = synth.make_lf(5, 5, 48, 48)
    clean, noisy = Hh.noisy_lf(lf, 25.0)
    P = O.make_params(25.0, 2.7, 4, 6, 2, 8, 4, "id", "sadct", "haar")
    n1, basic, st = O.run_step1(P, noisy, np.ones(25, np.uint32), O.ROWMAJOR, 5, 5, 1, 48, 48, 3)
    assert st.windows == 5 and st.passes == 5
    assert O.psnr_lf(basic, clean) > O.psnr_lf(noisy, clean) + 5


def test_sharded_passes_sum_to_the_full_pass():
    """Reference-patch rows are independent units with additive outputs: two half passes into zero
    buffers sum to the full pass (what the RCCL all-reduce relies on)."""
    lf = Hh.source_lf(crop=64)
    clean, noisy = Hh.noisy_lf(lf, 25.0)
    pk = (4, 6, 2, 8, 3, "id", "sadct", "haar")
    win, Wb, Hb = Hh.padded_window(noisy, 64, 64, 3, 8)
    num, den, st = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3)
    n_rows = int(round(np.sqrt(st.groups)))
    from lfbm5d_amd import core
    b0, e0 = core.shard_rows(n_rows, 0, 2)
    b1, e1 = core.shard_rows(n_rows, 1, 2)
    assert (b0, e1) == (0, n_rows) and e0 == b1
    na, da, _ = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3, rows=(b0, e0))
    nb, db, _ = Hh.oracle_pass(1, 25.0, pk, win, None, Wb, Hb, 3, rows=(b1, e1))
    np.testing.assert_allclose(na + nb, num, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(da + db, den, rtol=1e-5, atol=1e-6)


def test_reference_tile_mode_costs_half_a_db():
    """The reference's OpenMP mode (bm5d.cpp:411-708: every SAI cut into nb_threads tiles with a halo whose output is
    discarded) restated in the oracle: on the README command with 8 tiles it loses about 0.5 dB against the untiled
    result, as the stock CLI did in the survey's probe run (33.72 / 35.21 dB with its own noise
    draw against 34.20 / 35.72 untiled)."""
    clean, noisy = Hh.noisy_lf(Hh.source_lf(), 25.0)
    mask = np.ones(9, np.uint32)
    lib = O.lib()
    lib.orc_set_tiles(8)
    try:
        n1, b, st1 = O.run_step1(O.make_params(25.0, 2.7, *Hh.README_HT), noisy.copy(), mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
        _, _, d, st2 = O.run_step2(O.make_params(25.0, 2.7, *Hh.README_WIEN), n1.copy(), b.copy(), mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
    finally:
        lib.orc_set_tiles(1)
    pb, pd = O.psnr_lf(b, clean), O.psnr_lf(d, clean)
    assert 33.5 < pb < 33.85 and 35.0 < pd < 35.35, (pb, pd)          # untiled: 34.2073 / 35.7082
    assert st1.windows == st2.windows == 1 and st1.passes == 1
