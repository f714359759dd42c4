"""The CPU oracle (oracle/) against the golden .las files the REAL reference produced
(tests/golden/make_golden.py).  This is what pins the oracle: every option family and every
branch family of SURVEY.md App. E (multi-alignment pairs, Fusion, Bridge)."""
import os

import pytest

from conftest import ROOT, golden_cases, read_case, run_cli, compare_las

# cases whose read pairs need the Bridge realignment, not restated yet (host/bridge.c)
NEEDS_BRIDGE = set()


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference_las(built, tmp_path, name):
    case = read_case(name)
    if name in NEEDS_BRIDGE:
        pytest.xfail("Bridge (filter.c:1456-1571 + Compute_Alignment) is not built yet")
    run_cli(os.path.join(ROOT, "oracle", "oracle_daligner"), case, str(tmp_path))
    assert compare_las(case, str(tmp_path)) == []


def test_golden_cases_present():
    names = golden_cases()
    for need in ("tiny2", "tiny_j1", "tiny_s", "tiny_t", "tiny_I", "tiny_A", "tiny_k12", "indel", "noisy", "tandem", "fusion", "fusion2"):
        assert need in names


@pytest.mark.parametrize("name,what", [("tandem", "bridges"), ("fusion", "fusions"), ("fusion2", "fusions"),
                                       ("indel", "redundancy calls")])
def test_fixtures_reach_the_rare_branches(built, tmp_path, name, what):
    """The golden cases really enter Handle_Redundancies / Fusion / Bridge."""
    import re
    import subprocess
    from conftest import link_db
    case = read_case(name)
    link_db(case["dbdir"], str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "oracle", "oracle_daligner"), "-v"] + case["opts"] + ["G.1", "G.1"],
                         cwd=str(tmp_path), check=True, stdout=subprocess.PIPE, text=True).stdout
    m = re.search(r"redundancy calls (\d+) fusions (\d+) bridges (\d+)", out)
    got = dict(zip(["redundancy calls", "fusions", "bridges"], map(int, m.groups())))
    assert got[what] > 0, got
