"""The multi-rank exchange of the window graph between real PROCESSES on one GPU (round 5): RCCL refuses two ranks on one device, so
until now the multi-process issue order had only been replayed on the host.  lfbm5d_comm_init_ipc gives run_graph a second
transport -- device copies out of IPC-mapped peer buffers gated by words in mapped memory -- under the SAME issue order, event
gating, channels and abort path; tools/ipc_ranks.py starts fresh child processes and compares every rank's light fields with one
rank's, bit for bit; a peer that leaves must end the others with an error inside the watchdog, never with a hang."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, limit=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ipc_ranks.py")] + list(args), capture_output=True, text=True, timeout=limit)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, (r.stdout[-1500:], r.stderr[-1500:])
    return r.returncode, json.loads(line[-1])


@pytest.mark.parametrize("world,case", [(2, "5x5"), (2, "7x9"), (3, "7x9"), (2, "17x17x96")])
def test_processes_on_one_gpu_are_bit_identical_to_one_rank(world, case):
    rc, d = _run("run", str(world), case, "30")
    assert rc == 0 and d["ok"], d
    assert all(r["identical_to_single_rank"] and r["second_job_identical"] for r in d["ranks"]), d
    assert d["windows_over_ranks"] == d["single_rank_windows"]
    if case != "5x5":
        assert all(r["messages"] > 0 for r in d["ranks"]), d


@pytest.mark.parametrize("world,S,case", [(4, 2, "7x9"), (2, 2, "5x5"), (3, 3, "17x17x96")])
def test_spatial_bands_between_processes_match_the_emulated_teams(world, S, case):
    """Round 6: option spatial_bands between real processes -- S teams of world / S processes, a team's job on the IPC transport
    under the team's numbering, the stitch through IPC handles of every rank's packed chunk.  Bit for bit the result of the same
    teams played by emulated ranks on one context (which tests/test_gpu_denoise.py ties to the one-rank job on each band's crop)."""
    rc, d = _run(f"bands{S}", str(world), case, "30")
    assert rc == 0 and d["ok"], d
    assert all(r["identical_to_single_rank"] and r["second_job_identical"] for r in d["ranks"]), d


def test_a_peer_that_leaves_ends_the_job_with_an_error_not_a_hang():
    rc, d = _run("die", "2", "7x9", "8")
    assert rc == 0 and d["ok"], d
    assert d["exit_codes"][0] == 7 and "error" in d["ranks"][0] and d["ranks"][0]["seconds"] < 60, d
