"""bench.py's one-line contract and __graft_entry__.smoke() on the GPU box (small workload: configs[0]'s light field shape)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_names_every_baseline_configuration():
    """The workloads bench.py can time cover BASELINE.json's configurations; the default is the headline one."""
    sys.path.insert(0, ROOT)
    import bench
    names = set(bench.WORKLOADS)
    for w in ("lf3x3x256x256_sigma25", "lf3x3x256x256_sigma25_dct", "lf9x9x512x512_sigma25", "lf17x17x512x512_sigma25",
              "lf17x17x512x512_sigma10_bior", "lf15x15x625x434_sigma50_n1"):
        assert w in names
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="lf17x17x512x512_sigma25"' in src


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--workload", "lf3x3x256x256_sigma25"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "SAI-megapixels/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "lf3x3x256x256_sigma25" and "model" not in d["config"]
    # value = the units of the timed steps over the timed interval
    mp = 3 * 3 * 256 * 256 / 1e6
    assert abs(d["value"] - mp / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "per_step"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # round 6: the distance to this design's own floor next to the (saturated) contract fraction, and the per-step fractions up front
    for k in ("frac_ht", "frac_wiener", "design_bytes", "floor_ms", "x_over_floor", "x_over_floor_by_class"):
        assert k in r, k
    assert r["design_bytes"] > 0 and r["floor_ms"] > 0 and abs(r["x_over_floor"] - r["avg_launch_ms"] / r["floor_ms"]) < 1e-6 * r["x_over_floor"]
    for step in ("ht", "wiener"):
        p = r["per_step"][step]
        assert p["launches"] >= 1 and p["avg_launch_ms"] > 0
        assert p["frac"] <= 1.0 or "flag" in p               # a fraction above 1 is never printed unflagged
        assert abs(p["design_bytes"] - sum(v["design_bytes"] for v in p["floor_by_class"].values())) < 1.0
        assert p["filt_bytes"] > 0 and p["x_over_floor"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and d["value"] > c["value"]
    assert d["psnr"]["denoised"] > d["psnr"]["basic"] > d["psnr"]["noisy"]


@pytest.mark.gpu
def test_smoke_entry_point():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stderr[-2000:]
