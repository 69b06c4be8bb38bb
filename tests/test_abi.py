"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/lfbm5d.h declares, and fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
import os
import re

import pytest

import lfbm5d_amd as L
from lfbm5d_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lfbm5d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lfbm5d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(core.library_path())
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lfbm5d.h but not exported"


def test_struct_layouts_match_header():
    assert C.sizeof(core.Params) == 12 * 4
    assert C.sizeof(core.Stats) == 5 * 8 + 6 * 8 + 4 * 8


def test_shard_rows_partition():
    for n in (1, 7, 61, 125, 127):
        for world in (1, 2, 3, 8):
            spans = [core.shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.LfBm5dError, match="no HIP device"):
        L.Context(0)
    import numpy as np
    with pytest.raises(L.LfBm5dError):
        a = np.zeros((9, 3 * 16 * 16), np.float32)
        L.run_bm5d_1st_step(25.0, 2.7, a, np.ones(9, np.uint32), a.copy(), L.ROWMAJOR, 3, 3, 1, 16, 16, 3,
                            8, 18, 6, 16, 4, False, L.ID, L.SADCT, L.HAAR, L.OPP, 1)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "lfbm5d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the CPU oracle", "").replace("oracle/", "") or f == "__none__", f


def test_library_reads_the_environment_in_one_place():
    """Round 6: the run-time knobs are per-context options (lfbm5d_set_option); the library's only getenv is options_from_env
    (lfbm5d_options.h, called once by lfbm5d_create).  The drop-in and the CLI keep the program-level variables they read at start-up."""
    csrc = os.path.join(ROOT, "lfbm5d_amd", "csrc")
    sites = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            n = len(re.findall(r"\bgetenv\s*\(", re.sub(r"/\*.*?\*/|//[^\n]*", "", open(os.path.join(csrc, f)).read(), flags=re.S)))
            if n:
                sites[f] = n
    assert sites.get("lfbm5d_options.h") == 1
    assert set(sites) <= {"lfbm5d_options.h", "run_bm5d.cpp", "lfbm5d_cli.cpp"}, sites
    assert sum(sites.values()) <= 8, sites
    # every option key has its old variable as an alias, and the Python mirror hands exactly those variables over
    src = open(os.path.join(csrc, "lfbm5d_options.h")).read()
    envs = set(re.findall(r'"(LFBM5D_[A-Z0-9_]+)"', src))
    assert envs == set(core.OPTION_ENV)


@pytest.mark.gpu
def test_options_are_per_context():
    a, b = L.Context(0), L.Context(0)
    try:
        assert a.get_option("lanes") == os.environ.get("LFBM5D_LANES", "2")
        a.set_option("lanes", 3)
        a.set_option("LFBM5D_STEP_SHARDING", "rows")
        a.set_option("dct8w_v2", 1)
        assert (a.get_option("lanes"), a.get_option("step_sharding"), a.get_option("dct8w_v2")) == ("3", "rows", "1")
        assert (b.get_option("lanes"), b.get_option("step_sharding"), b.get_option("dct8w_v2")) == (os.environ.get("LFBM5D_LANES", "2"), "0", "0")
        a.set_option("lanes", None)
        assert a.get_option("lanes") == "2"
        # round 6's keys: spatial bands (0 = the library's rule), the halo, the filt layout hook
        assert (a.get_option("spatial_bands"), a.get_option("band_halo"), a.get_option("filt_group_major")) == ("1", "0", "0")
        a.set_option("spatial_bands", 0); a.set_option("band_halo", 48); a.set_option("LFBM5D_FILT_GROUP_MAJOR", "1")
        assert (a.get_option("spatial_bands"), a.get_option("band_halo"), a.get_option("filt_group_major")) == ("0", "48", "1")
        assert (b.get_option("spatial_bands"), b.get_option("band_halo"), b.get_option("filt_group_major")) == ("1", "0", "0")
        with pytest.raises(L.LfBm5dError, match="unknown option"):
            a.set_option("no_such_option", 1)
    finally:
        a.close(); b.close()
