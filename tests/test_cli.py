"""The drop-in command line: PNG codec self-check on CPU, the README test command on the GPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "lfbm5d_amd", "LFBM5Ddenoising")


def write_source_lf(tmp):
    from PIL import Image
    lf = np.load(os.path.join(ROOT, "tests", "golden", "sourceLF_3x3_256_u8.npy"))
    src = os.path.join(tmp, "sourceLF")
    os.makedirs(src)
    for s in range(3):
        for t in range(3):
            Image.fromarray(lf[s * 3 + t].transpose(1, 2, 0)).save(f"{src}/SAI_{s + 1:02d}_{t + 1:02d}.png")
    return src, lf


def test_png_codec_round_trip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for shape, mode in (((37, 53, 3), "RGB"), ((20, 31), "L"), ((16, 16, 4), "RGBA")):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        src, dst = str(tmp_path / f"in_{mode}.png"), str(tmp_path / f"out_{mode}.png")
        Image.fromarray(a, mode).save(src)   # PIL picks per-row filters: exercises all unfilter paths
        subprocess.check_call([CLI, "--png-roundtrip", src, dst], stdout=subprocess.DEVNULL)
        b = np.asarray(Image.open(dst))
        assert np.array_equal(b, a[..., :3] if mode == "RGBA" else a)


def test_usage_on_missing_arguments():
    r = subprocess.run([CLI, "a", "b"], capture_output=True, text=True)
    assert r.returncode != 0 and "usage:" in r.stdout


@pytest.mark.gpu
def test_readme_command_matches_oracle(tmp_path):
    """README.md:50 test command with LFBM5D_SEED=1: same noise as the oracle harness, PSNR report
    within the north-star tolerance of the oracle's (and of the reference-run numbers of SURVEY 6)."""
    tmp = str(tmp_path)
    src, lf = write_source_lf(tmp)
    for d in ("noisy", "basic", "denoised", "diff"):
        os.makedirs(os.path.join(tmp, d))
    res = os.path.join(tmp, "measures.txt")
    args = [CLI, src, "SAI", "_", "3", "3", "1", "1", "1", "1", "row", "25", "2.7", f"{tmp}/noisy", f"{tmp}/basic",
            f"{tmp}/denoised", f"{tmp}/diff", "8", "18", "6", "16", "4", "id", "sadct", "haar", "0", "16", "18", "6", "8", "4",
            "dct", "sadct", "haar", "0", "opp", "0", res]
    out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, LFBM5D_SEED="1"))
    assert out.returncode == 0, out.stdout[-2000:]
    txt = open(res).read()
    vals = {k: float(txt.split(f"-> Average PSNR {k} = ")[1].split()[0]) for k in ("noisy", "basic", "denoised")}
    assert abs(vals["noisy"] - 20.1672) < 1e-3
    assert abs(vals["basic"] - 34.2073) < 0.01 and abs(vals["denoised"] - 35.7082) < 0.01
    from PIL import Image
    im = np.asarray(Image.open(f"{tmp}/denoised/SAI_02_02.png")).astype(np.float32).transpose(2, 0, 1)
    mse = ((im - lf[4].astype(np.float32)) ** 2).mean()
    assert 20 * np.log10(255 / np.sqrt(mse)) > 36.0


@pytest.mark.gpu
def test_readme_command_as_one_job_writes_the_same_files(tmp_path):
    """LFBM5D_ONE_JOB=1: the CLI calls run_bm5d() (both steps as one dependency graph of windows, lfbm5d_denoise_host) instead of
    run_bm5d_1st_step + run_bm5d_2nd_step: the denoised PNGs and PSNR are the same, byte for byte.  The basic estimate is the
    one the two calls leave at the END -- after the second step's forward + inverse colour transform of LF_basic (bm5d.cpp:829,
    :1416: the reference's matrices are not inverses) -- where the two-step CLI reports and saves it in between:
    0.015 dB apart on this light field."""
    outs = {}
    for mode in ("two", "one"):
        tmp = os.path.join(str(tmp_path), mode)
        os.makedirs(tmp)
        src, lf = write_source_lf(tmp)
        for d in ("noisy", "basic", "denoised", "diff"):
            os.makedirs(os.path.join(tmp, d))
        res = os.path.join(tmp, "measures.txt")
        args = [CLI, src, "SAI", "_", "3", "3", "1", "1", "1", "1", "row", "25", "2.7", f"{tmp}/noisy", f"{tmp}/basic",
                f"{tmp}/denoised", f"{tmp}/diff", "8", "18", "6", "16", "4", "id", "sadct", "haar", "0", "16", "18", "6", "8", "4",
                "dct", "sadct", "haar", "0", "opp", "0", res]
        env = dict(os.environ, LFBM5D_SEED="1")
        if mode == "one":
            env["LFBM5D_ONE_JOB"] = "1"
        out = subprocess.run(args, capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout[-2000:]
        assert ("Steps 1 and 2 done in" in out.stdout) == (mode == "one")
        txt = open(res).read()
        outs[mode] = ({k: txt.split(f"-> Average PSNR {k} = ")[1].split()[0] for k in ("noisy", "basic", "denoised")},
                      {f"{d}/SAI_0{s}_0{t}.png": open(f"{tmp}/{d}/SAI_0{s}_0{t}.png", "rb").read()
                       for d in ("basic", "denoised") for s in (1, 2, 3) for t in (1, 2, 3)})
    assert outs["one"][0]["noisy"] == outs["two"][0]["noisy"] and outs["one"][0]["denoised"] == outs["two"][0]["denoised"]
    assert 0 < abs(float(outs["one"][0]["basic"]) - float(outs["two"][0]["basic"])) < 0.05
    for name, png in outs["two"][1].items():
        if name.startswith("denoised/"):
            assert outs["one"][1][name] == png, name


@pytest.mark.gpu
def test_readme_command_in_tile_mode_matches_the_oracles_tile_mode(tmp_path):
    """nbThreads = 8 with LFBM5D_TILED=1: the drop-in reproduces the reference's OpenMP tile mode (bm5d.cpp:411-708) --
    PSNR report within 0.01 dB of the oracle's tiled run on the same noise, half a dB below the untiled numbers."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as Hh
    from oracle import oracle as O
    tmp = str(tmp_path)
    src, lf = write_source_lf(tmp)
    for d in ("noisy", "basic", "denoised", "diff"):
        os.makedirs(os.path.join(tmp, d))
    res = os.path.join(tmp, "measures.txt")
    args = [CLI, src, "SAI", "_", "3", "3", "1", "1", "1", "1", "row", "25", "2.7", f"{tmp}/noisy", f"{tmp}/basic",
            f"{tmp}/denoised", f"{tmp}/diff", "8", "18", "6", "16", "4", "id", "sadct", "haar", "0", "16", "18", "6", "8", "4",
            "dct", "sadct", "haar", "0", "opp", "8", res]
    out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, LFBM5D_SEED="1", LFBM5D_TILED="1"))
    assert out.returncode == 0, out.stdout[-2000:]
    txt = open(res).read()
    vals = {k: float(txt.split(f"-> Average PSNR {k} = ")[1].split()[0]) for k in ("noisy", "basic", "denoised")}
    clean, noisy = Hh.noisy_lf(Hh.source_lf(), 25.0)
    mask = np.ones(9, np.uint32)
    lib = O.lib()
    lib.orc_set_tiles(8)
    try:
        n1, b, _ = O.run_step1(O.make_params(25.0, 2.7, *Hh.README_HT), noisy.copy(), mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
        _, _, d, _ = O.run_step2(O.make_params(25.0, 2.7, *Hh.README_WIEN), n1.copy(), b.copy(), mask, O.ROWMAJOR, 3, 3, 1, 256, 256, 3)
    finally:
        lib.orc_set_tiles(1)
    assert abs(vals["noisy"] - 20.1672) < 1e-3
    assert abs(vals["basic"] - O.psnr_lf(b, clean)) < 0.01 and abs(vals["denoised"] - O.psnr_lf(d, clean)) < 0.01
    assert vals["basic"] < 34.2073 - 0.3 and vals["denoised"] < 35.7082 - 0.3      # the untiled result (test above)


CLI3 = os.path.join(ROOT, "lfbm5d_amd", "LFBM3Ddenoising")


def test_bm3d_usage_on_missing_arguments():
    r = subprocess.run([CLI3, "a", "b"], capture_output=True, text=True)
    assert r.returncode != 0 and "usage:" in r.stdout


@pytest.mark.gpu
def test_bm3d_readme_command_matches_oracle(tmp_path):
    """README.md:51 test command (LFBM3Ddenoising, BM3D on every SAI) with LFBM5D_SEED=1 against the oracle's
    restatement of run_bm3d_LF on the same noise; four SAIs of the light field to bound the oracle's time."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    tmp = str(tmp_path)
    src, lf = write_source_lf(tmp)
    for d in ("noisy", "basic", "denoised", "diff"):
        os.makedirs(os.path.join(tmp, d))
    res = os.path.join(tmp, "measures3d.txt")
    args = [CLI3, src, "SAI", "_", "2", "2", "1", "1", "1", "1", "row", "25", "2.7", f"{tmp}/noisy", f"{tmp}/basic",
            f"{tmp}/denoised", f"{tmp}/diff", "16", "16", "8", "3", "bior", "0", "32", "16", "8", "3", "dct", "0", "opp", "8", res]
    out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, LFBM5D_SEED="1"))
    assert out.returncode == 0, out.stdout[-2000:]
    txt = open(res).read()
    vals = {k: float(txt.split(f"-> Average PSNR {k} = ")[1].split()[0]) for k in ("noisy", "basic", "denoised")}
    sel = [0, 1, 3, 4]                                      # SAI_01_01, _01_02, _02_01, _02_02 in the 3x3 golden array
    clean = np.ascontiguousarray(lf[sel].astype(np.float32)).reshape(4, -1)
    noisy = O.add_noise_lf(clean, 25.0, seed=1)
    _, b_o, d_o, _ = O.run_bm3d_lf(25.0, 2.7, noisy, np.ones(4, np.uint32), 256, 256, 3, (16, 16, 8, 3, "bior", 0), (32, 16, 8, 3, "dct", 0))
    assert abs(vals["noisy"] - O.psnr_lf(noisy, clean)) < 1e-3
    assert abs(vals["basic"] - O.psnr_lf(b_o, clean)) < 0.01 and abs(vals["denoised"] - O.psnr_lf(d_o, clean)) < 0.01
    assert vals["denoised"] > vals["basic"] > vals["noisy"] + 10
    from PIL import Image
    im = np.asarray(Image.open(f"{tmp}/denoised/SAI_02_02.png")).astype(np.float32).transpose(2, 0, 1)
    assert np.abs(im - np.clip(np.round(d_o[3].reshape(3, 256, 256)), 0, 255)).max() <= 1
