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


@pytest.mark.parametrize("name", ["tandem", "tandem2", "tan_O", "fusion", "fusion2", "indel", "noisy", "prod", "tiny_s"])
def test_product_host_tail_behind_the_oracle_front_matches_reference_las(built, tmp_path, name):
    """The PRODUCT's bridge realignment (damar_amd/csrc/host/bridge.c, written from the algorithm; the oracle has its own
    reference-shaped oracle/bridge.c) behind the oracle's CPU front: the cases with Bridge / Fusion / multi-record pairs
    against the reference's files, where there is no GPU (oracle/oracle_daligner_hosttail, oracle/Makefile)."""
    exe = os.path.join(ROOT, "oracle", "oracle_daligner_hosttail")
    case = read_case(name)
    if case.get("tool", "daligner") != "daligner":
        pytest.skip("a datander case: no bridges there (scrub/tandem.c:767-850)")
    run_cli(exe, case, str(tmp_path))
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


def test_oracle_random_option_combinations_equal_reference(built, tmp_path):
    """The same kind of sweep one level down: the oracle against the REAL reference binary
    (oracle/_ref/daligner, present wherever /root/reference was at build time) over random
    combinations of options, so that the oracle is pinned across the option space and not only on
    the per-family golden cases."""
    import random
    import subprocess
    from conftest import GOLDEN, ROOT, link_db
    ref = os.path.join(ROOT, "oracle", "_ref", "daligner")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/daligner not built")
    rng = random.Random(77)
    # three databases: two blocks with mask tracks, one block of reads with tandem arrays (many seeds per pair, Bridge and
    # Fusion in the host tail), one block of long reads (wide bands); 32 combinations in all
    for n in range(32):
        gold = ("mask_dust", "tandem", "long", "mask_dust")[n % 4]
        dbdir = os.path.join(GOLDEN, gold)
        # (k = 10 on the repeat-rich reads costs the reference minutes: 13 700 mutual matches per k-mer pair)
        opts = ["-k%d" % rng.choice([10, 12, 14, 16, 18] if gold == "mask_dust" else [12, 14, 16, 18]),
                "-w%d" % rng.choice([4, 5, 6, 7]), "-h%d" % rng.choice([25, 35, 50]),
                "-e%g" % rng.choice([.65, .7, .8]), "-l%d" % rng.choice([500, 1000, 2000]), "-s%d" % rng.choice([50, 100, 126, 200]),
                "-j%d" % rng.choice([1, 2, 4, 8])]
        t = rng.choice([0, 0, 8, 20])
        opts += (["-t%d" % t] if t else []) + (["-I"] if rng.random() < .5 else []) + (["-A"] if rng.random() < .3 else [])
        if gold == "mask_dust":
            opts += rng.choice([[], ["-mdust"], ["-mdust", "-mrnd"]])
        opts += ["-b"] if rng.random() < .3 else []
        blocks = ["G.2", "G.2", "G.1"] if gold == "mask_dust" else ["G.1", "G.1"]
        rdir, odir = os.path.join(str(tmp_path), "r%d" % n), os.path.join(str(tmp_path), "o%d" % n)
        link_db(dbdir, rdir)
        link_db(dbdir, odir)
        for exe, d in ((ref, rdir), (os.path.join(ROOT, "oracle", "oracle_daligner"), odir)):
            subprocess.run([exe] + opts + blocks, cwd=d, check=True, stdout=subprocess.DEVNULL)
        nlas = 0
        for dp, _, fs in os.walk(rdir):
            for f in fs:
                if f.endswith(".las"):
                    rel = os.path.relpath(os.path.join(dp, f), rdir)
                    assert open(os.path.join(dp, f), "rb").read() == open(os.path.join(odir, rel), "rb").read(), (gold, opts, rel)
                    nlas += 1
        assert nlas >= (2 if gold == "mask_dust" else 1), (gold, opts)
