import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces must exist (build() makes them); tests never rebuild silently."""
    need = ["damar_amd/libdamar_hip.so", "damar_amd/bin/daligner", "damar_amd/bin/simdb",
            "oracle/liboracle.so", "oracle/oracle_daligner", "oracle/oracle_lastrace"]
    missing = [f for f in need if not os.path.exists(os.path.join(ROOT, f))]
    if missing:
        import __graft_entry__ as g
        g.build()
    return ROOT


def golden_cases():
    out = []
    for name in sorted(os.listdir(GOLDEN)):
        if os.path.exists(os.path.join(GOLDEN, name, "case.txt")):
            out.append(name)
    return out


def read_case(name):
    d = os.path.join(GOLDEN, name)
    db, opts, lines, tool = None, [], [], "daligner"
    for ln in open(os.path.join(d, "case.txt")):
        w = ln.split()
        if w[0] == "db":
            db = w[1]
        elif w[0] == "opts":
            opts = w[1:]
        elif w[0] == "line":
            lines.append((w[1], w[2:]))
        elif w[0] == "tool":
            tool = w[1]
    las, md5s = [], {}
    for dp, _, fs in os.walk(os.path.join(d, "las")):
        for f in fs:
            las.append(os.path.relpath(os.path.join(dp, f), os.path.join(d, "las")))
    if os.path.exists(os.path.join(d, "las.md5")):          # a case whose files are too large to keep: their md5s
        for ln in open(os.path.join(d, "las.md5")):
            m, rel = ln.split()
            md5s[rel] = m
    return dict(name=name, dbdir=os.path.join(GOLDEN, db), opts=opts, lines=lines, tool=tool,
                lasdir=os.path.join(d, "las"), las=sorted(las), las_md5=md5s)


def link_db(dbdir, dst, root="G"):
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(dbdir)):
        if f == "%s.db" % root or f.startswith(".%s." % root):          # stub, .idx, .bps and track files
            os.symlink(os.path.join(dbdir, f), os.path.join(dst, f))


def run_cli(exe, case, workdir, env=None):
    """exe: the daligner-like binary; datander cases use its sibling *datander binary."""
    link_db(case["dbdir"], workdir)
    if case["tool"] == "datander":
        exe = exe.replace("daligner", "datander")
    for a, bs in case["lines"]:
        subprocess.run([exe] + case["opts"] + ["G." + a] + ["G." + b for b in bs], cwd=workdir, check=True,
                       stdout=subprocess.DEVNULL, env=env)


def opts_to_plan_kwargs(opts):
    kw = {}
    for o in opts:
        f, v = o[1], o[2:]
        if f in "kwhtlsj":
            kw[f] = int(v)
        elif f == "r":
            kw["run"] = int(v)
        elif f == "e":
            kw["e"] = float(v)
        elif f == "I":
            kw["identity"] = 1
        elif f == "A":
            kw["symmetric"] = 0
        elif f == "m":
            kw.setdefault("masks", []).append(v)
        elif f == "b":
            kw["biased"] = 1
        elif f == "O":                  # daligner.c:737-739: -O implies -I
            kw["only_identity"] = 1
            kw["identity"] = 1
        elif f == "T":
            kw["no_trace"] = 1
    return kw


def compare_las(case, workdir):
    bad = []
    for rel in case["las"]:
        a = os.path.join(case["lasdir"], rel)
        b = os.path.join(workdir, rel)
        if not os.path.exists(b) or open(a, "rb").read() != open(b, "rb").read():
            bad.append(rel)
    import hashlib
    for rel, m in case.get("las_md5", {}).items():
        b = os.path.join(workdir, rel)
        if not os.path.exists(b) or hashlib.md5(open(b, "rb").read()).hexdigest() != m:
            bad.append(rel)
    return bad
