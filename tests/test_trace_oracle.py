"""(f)4 Compute_Trace_PTS: the oracle's restatement (oracle/trace.c) against the REAL reference.

tests/golden/trace_ref_md5.txt holds the md5 of what the reference's Compute_Trace_PTS (align.c:5577,
driven by oracle/ref_lastrace.c exactly like utils/LAshow.c:245-262) leaves for every record of every
golden .las in the three modes LOWERMOST / GREEDIEST / UPPERMOST (made by `make_golden.py trace`)."""
import hashlib
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, read_case


def ref_lines(fixture="trace_ref_md5.txt"):
    out = []
    for ln in open(os.path.join(GOLDEN, fixture)):
        md5, name, las, mode = ln.split()
        out.append((md5, name, las, int(mode)))
    return out


@pytest.mark.parametrize("fixture,extra", [("trace_ref_md5.txt", []), ("trace_mid_ref_md5.txt", ["mid"])])
def test_oracle_trace_equals_reference_on_golden(built, tmp_path, fixture, extra):
    """Compute_Trace_PTS (align.c:5577) and Compute_Trace_MID (align.c:5694), every golden record, three modes."""
    from concurrent.futures import ThreadPoolExecutor
    tool = os.path.join(ROOT, "oracle", "oracle_lastrace")

    def one(job):
        i, (md5, name, las, mode) = job
        c = read_case(name)
        out = str(tmp_path / ("o%d.bin" % i))
        subprocess.run([tool, os.path.join(c["dbdir"], "G"), os.path.join(c["lasdir"], las), out, str(mode)] + extra, check=True)
        got = hashlib.md5(open(out, "rb").read()).hexdigest()
        os.unlink(out)
        return got == md5, (name, las, mode, extra)

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
        res = list(pool.map(one, enumerate(ref_lines(fixture))))
    bad = [what for ok, what in res if not ok]
    assert not bad, bad
    assert len(res) >= 100


def test_oracle_trace_equals_reference_live(built, tmp_path):
    """Where the reference build is present (this container, or oracle/_ref carried to the GPU box): a case
    outside the fixture list, 16-bit trace points (-s126 style spacing is covered by tiny_s in the list)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_lastrace")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built")
    c = read_case("noisy")
    for las in c["las"]:
        for mode in (0, 1, -1):
            a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
            args = [os.path.join(c["dbdir"], "G"), os.path.join(c["lasdir"], las)]
            for extra in ([], ["mid"]):
                subprocess.run([ref] + args + [a, str(mode)] + extra, check=True)
                subprocess.run([os.path.join(ROOT, "oracle", "oracle_lastrace")] + args + [b, str(mode)] + extra, check=True)
                assert open(a, "rb").read() == open(b, "rb").read()
