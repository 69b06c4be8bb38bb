"""The two generations of the block-matching scan (lfbm5d_bm.hip: one wave per table; lfbm5d_scan2.hip: ring-sharing
workgroups) must produce the same bits: raw disparity tables, self-search scores, selections and the pass's sums, on
regular grids (pattern stores), on grids whose forced last row / column is off the pattern (look-up stores, p = 3), and on
search windows other than the README's (precompute_BM core:3301-3461, precompute_BM_stereo core:3479-3611)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "ht-k16-128": ["128", "128", "1"],
    "wiener-k8-ragged": ["96", "150", "2"],
    "ht-k16-p3-offpattern": ["101", "131", "1", "25", "3"],
    "wiener-k8-p3-offpattern": ["99", "122", "2", "25", "3"],
    "ht-k16-ndisp3-nsim9": ["120", "90", "1", "10", "4", "3", "9"],
    "wiener-k8-ndisp8-nsim12-p5": ["110", "140", "2", "50", "5", "8", "12"],
}


@pytest.mark.parametrize("name", list(CASES))
def test_scan_generations_agree_bit_for_bit(name):
    env = dict(os.environ)
    env.pop("LFBM5D_SCAN_V1", None); env.pop("LFBM5D_SCAN_FULL_TABLES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_ab.py")] + CASES[name], capture_output=True, text=True, env=env, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stderr[-2000:]
    d = json.loads(line[-1])
    assert d["versions"] == [1, 2], d
    for k in ("tables_differ", "scores_differ", "self_idx_differ", "self_cnt_differ", "best_differ", "shape_differ"):
        assert d[k] == 0, (k, d)
    assert d["num_equal"] and d["den_equal"], d
    # ... and the default, combined form (tables reduced inside the workgroup): same selections, same sums
    assert d["combined_version"] == 3, d
    for k in ("best_differ", "shape_differ", "self_idx_differ", "self_cnt_differ", "scores_differ"):
        assert d["combined"][k] == 0, (k, d["combined"])
    assert d["combined"]["num_equal"] and d["combined"]["den_equal"], d["combined"]
    assert r.returncode == 0
