"""GPU (CLI binary) vs oracle_daligner for explicit option strings on a fixture DB block pair."""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, link_db
name = sys.argv[1]
for optstr in sys.argv[2:]:
    opts = optstr.split()
    w = tempfile.mkdtemp(dir="/dev/shm")
    res = []
    for sub, exe in (("g", os.path.join(ROOT, "damar_amd", "bin", "daligner")), ("o", os.path.join(ROOT, "oracle", "oracle_daligner"))):
        d = os.path.join(w, sub); link_db(os.path.join(GOLDEN, name), d)
        subprocess.run([exe] + opts + ["G.1", "G.1"], cwd=d, check=True, stdout=subprocess.DEVNULL)
        res.append(open(os.path.join(d, "d001_00001", "G.1.G.1.las"), "rb").read())
    print(optstr, "->", "same" if res[0] == res[1] else "DIFF (%d vs %d bytes)" % (len(res[0]), len(res[1])), flush=True)
    shutil.rmtree(w)
