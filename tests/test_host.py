"""Host-side pieces of the path: synthetic DB writer, block loader, complement, .las I/O."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def test_simdb_is_deterministic_and_matches_golden_db(built, tmp_path):
    from damar_amd import api
    nb = api.sim_write_db(str(tmp_path), "G", 0.1, coverage=12., seed=11, block_mbp=1)
    assert nb == 2
    for f in ("G.db", ".G.bps"):
        assert md5(os.path.join(str(tmp_path), f)) == md5(os.path.join(GOLDEN, "tiny2", f))
    a = np.fromfile(os.path.join(str(tmp_path), ".G.idx"), dtype=np.uint8)
    b = np.fromfile(os.path.join(GOLDEN, "tiny2", ".G.idx"), dtype=np.uint8)
    assert np.array_equal(a, b)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "FA2db")), reason="needs oracle/_ref")
def test_simdb_equals_simulator_fa2db_dbsplit(built, tmp_path):
    """simulate->DB writer vs the reference's own simulator | FA2db ; DBsplit."""
    from damar_amd import api
    ref = os.path.join(ROOT, "oracle", "_ref")
    d1, d2 = str(tmp_path / "a"), str(tmp_path / "b")
    os.makedirs(d1)
    with open(os.path.join(d1, "sim.fasta"), "w") as f:
        subprocess.run([os.path.join(ref, "simulator"), "0.2", "-c10", "-r5", "-e.15"], stdout=f, check=True)
    subprocess.run([os.path.join(ref, "FA2db"), "G", "sim.fasta"], cwd=d1, check=True, stdout=subprocess.DEVNULL)
    subprocess.run([os.path.join(ref, "DBsplit"), "-s1", "G"], cwd=d1, check=True, stdout=subprocess.DEVNULL)
    api.sim_write_db(d2, "G", 0.2, coverage=10., seed=5, block_mbp=1)
    assert md5(os.path.join(d1, ".G.bps")) == md5(os.path.join(d2, ".G.bps"))
    assert open(os.path.join(d1, "G.db")).read() == open(os.path.join(d2, "G.db")).read()
    a = np.fromfile(os.path.join(d1, ".G.idx"), dtype=np.uint8).copy()
    b = np.fromfile(os.path.join(d2, ".G.idx"), dtype=np.uint8).copy()
    assert len(a) == len(b)
    for x in (a, b):                       # struct padding and dumped pointers
        x[44:48] = 0
        x[48:88] = 0
        r = x[88:].reshape(-1, 32)
        r[:, 4:8] = 0
        r[:, 28:32] = 0
    assert np.array_equal(a, b)


def test_read_block_layout(built):
    from damar_amd import api
    db = api.read_block(os.path.join(GOLDEN, "tiny2", "G.2"))
    assert db.part == 2 and db.nreads > 0 and db.ufirst > 0
    bases = (C.c_char * (db.reads[db.nreads].boff)).from_address(db.bases)
    assert C.c_char.from_address(db.bases - 1).value == b"\x04"       # db/DB.c:1570
    tot, mx = 0, 0
    for i in range(db.nreads):
        r = db.reads[i]
        assert bases[r.boff + r.rlen] == b"\x04"
        assert db.reads[i + 1].boff == r.boff + r.rlen + 1
        tot += r.rlen
        mx = max(mx, r.rlen)
    assert tot == db.totlen and mx == db.maxlen
    raw = np.frombuffer(bases, dtype=np.uint8)
    assert raw.max() == 4
    api.lib().damar_close_block(C.byref(db))


def test_complement_is_an_involution(built):
    from damar_amd import api
    L = api.lib()
    db = api.read_block(os.path.join(GOLDEN, "tiny2", "G.1"))
    n = db.reads[db.nreads].boff
    before = bytes((C.c_char * n).from_address(db.bases))
    f0 = list(db.freq)
    L.damar_complement_block(C.byref(db), 1)
    mid = bytes((C.c_char * n).from_address(db.bases))
    assert mid != before and list(db.freq) == [f0[3], f0[2], f0[1], f0[0]]
    r0 = db.reads[0]
    assert mid[r0.boff] == 3 - before[r0.boff + r0.rlen - 1]
    L.damar_complement_block(C.byref(db), 1)
    assert bytes((C.c_char * n).from_address(db.bases)) == before
    L.damar_close_block(C.byref(db))


def test_get_dir_and_las_stats(built):
    from damar_amd import api, driver
    assert api.get_dir(1, 1) == "d001_00001"
    assert api.get_dir(12, 345) == "d012_00345"
    assert api.get_dir(3, 0) == "."
    n, bp = driver.las_stats(os.path.join(GOLDEN, "tiny2", "las", "d001_00001", "G.1.G.1.las"))
    assert n > 100 and bp > 100000
    assert driver.hpc_plan(3) == [(1, [1]), (2, [2, 1]), (3, [3, 2, 1])]


def test_missing_library_fails_loudly(monkeypatch):
    from damar_amd import lib as dl
    monkeypatch.setattr(dl, "_LIB", None)
    monkeypatch.setattr(dl, "lib_path", lambda: "/nonexistent/libdamar_hip.so")
    with pytest.raises(dl.LibraryMissing):
        dl.load()


def test_complement_copy_equals_in_place_complement_with_masks(built):
    """damar_complement_copy (the re-entrant copy the command-line driver makes on its second thread) against
    the in-place complement_DB restatement, on a block with a merged mask track: bases, base frequencies and
    the mirrored mask intervals (daligner.c:511-628)."""
    from damar_amd import api
    L = api.lib()
    L.damar_complement_copy.argtypes = [C.POINTER(api.HITS_DB), C.POINTER(api.HITS_DB)]
    L.damar_free_complement.argtypes = [C.POINTER(api.HITS_DB)]
    L.damar_load_masks.argtypes = [C.POINTER(api.HITS_DB), C.POINTER(C.c_char_p), C.c_int]

    class Track(C.Structure):
        pass
    Track._fields_ = [("next", C.POINTER(Track)), ("name", C.c_char_p), ("size", C.c_int), ("anno", C.c_void_p), ("data", C.c_void_p)]

    def masks(db):
        t = C.cast(db.tracks, C.POINTER(Track))
        assert bool(t)
        n = db.nreads
        anno = np.ctypeslib.as_array(C.cast(t.contents.anno, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        data = np.ctypeslib.as_array(C.cast(t.contents.data, C.POINTER(C.c_int32)), shape=(max(int(anno[n]), 1),))[:int(anno[n])].copy()
        return anno, data
    name = os.path.join(ROOT, "tests", "golden", "mask_dust", "G.1").encode()
    a, b = api.HITS_DB(), api.HITS_DB()
    names = (C.c_char_p * 2)(b"dust", b"rnd")
    for db in (a, b):
        assert L.damar_read_block(name, C.byref(db)) == 0
        assert L.damar_load_masks(C.byref(db), names, 2) == 0
    cp = api.HITS_DB()
    L.damar_complement_copy(C.byref(a), C.byref(cp))
    L.damar_complement_block(C.byref(b), 1)
    tot = a.reads[a.nreads].boff
    got = np.ctypeslib.as_array(C.cast(cp.bases, C.POINTER(C.c_int8)), shape=(tot,))
    want = np.ctypeslib.as_array(C.cast(b.bases, C.POINTER(C.c_int8)), shape=(tot,))
    assert (got == want).all()
    assert list(cp.freq) == list(b.freq)
    ga, gd = masks(cp)
    wa, wd = masks(b)
    assert (ga == wa).all() and (gd == wd).all() and len(gd) > 100
    fwd = np.ctypeslib.as_array(C.cast(a.bases, C.POINTER(C.c_int8)), shape=(tot,))
    assert not (fwd == got).all()                      # the original is untouched and differs
    L.damar_free_complement(C.byref(cp))
    L.damar_close_block(C.byref(a))
    L.damar_close_block(C.byref(b))


def fasta_inputs(d):
    """Three FASTA files that exercise FA2db's header rules: PacBio headers with several reads per well,
    short reads that -x drops (their file index still counts), upper / lower case and N, long and short
    lines; headers that are not PacBio; a file whose first header is not PacBio but later ones are."""
    import random
    rng = random.Random(11)

    def seq(n, alphabet="acgt"):
        return "".join(rng.choice(alphabet) for _ in range(n))
    with open(os.path.join(d, "pb.fasta"), "w") as f:
        for i in range(40):
            n = rng.choice([300, 999, 1000, 1001, 2500, 7000, 12000])
            s = seq(n, "acgtACGTnN")
            f.write(">m140913_050931_42139_c1_s1_p0/%d/%d_%d RQ=0.85%d\n" % (i // 2 + 7, 10 * i, 10 * i + n, i % 10))
            w = rng.choice([60, 70, 80, 100000])
            for j in range(0, n, w):
                f.write(s[j:j + w] + "\n")
    with open(os.path.join(d, "plain.fa"), "w") as f:
        for i in range(15):
            f.write(">read_%d some text\n%s\n" % (i, seq(rng.choice([1500, 3000, 800]))))
    with open(os.path.join(d, "mixed.fasta"), "w") as f:
        for i in range(12):
            n = rng.choice([1200, 5000])
            f.write((">Sim/%d/0_%d RQ=0.850" % (i + 1, n) if i % 3 else ">odd/%d" % i) + "\n" + seq(n) + "\n")
    return ["pb.fasta", "plain.fa", "mixed.fasta"]


def fasta_header_inputs(d):
    """Two FASTA files whose headers carry arguments (NAME=v1,v2,...) for FA2db -c / -Q: a quality value with the "0."
    the reference strips, readType (some reads without it, some not FullHqRead), a list-valued argument, the chemistry
    string (stored as characters), an argument nobody asks for."""
    import random
    rng = random.Random(3)

    def seq(n):
        return "".join(rng.choice("acgt") for _ in range(n))
    with open(os.path.join(d, "h.fasta"), "w") as f:
        for i in range(30):
            n = rng.choice([900, 1500, 3000])
            args = ["RQ=0.8%02d" % i]
            if i % 3 != 2:
                args.append("readType=%d" % rng.choice([0, 1, 2, 3]))
            if i % 4 == 0:
                args.append("SN=%d,%d,%d" % (i, 2 * i, 3 * i))
            if i % 5 == 0:
                args.append("chemistry=P6C4")
            if i % 7 == 0:
                args.append("other=5")
            f.write(">m1/%d/%d_%d %s\n%s\n" % (i // 2 + 3, 10 * i, 10 * i + n, " ".join(args), seq(n)))
    with open(os.path.join(d, "g.fasta"), "w") as f:
        for i in range(10):
            f.write(">m2/%d/0_2000 readType=%d SN=7\n%s\n" % (i + 1, 1 + i % 2, seq(2000)))
    return ["h.fasta", "g.fasta"]


HEADER_TRACK_RUNS = {          # FA2db -c / -Q (FA2db.c:169-355, 638-643, 809-837)
    "c":      [["-x1000", "-cRQ", "-cSN", "-creadType", "-cchemistry", "T", "h.fasta", "g.fasta"]],
    "Q":      [["-x1000", "-b", "-Q", "-creadType", "-cchemistry", "T", "h.fasta", "g.fasta"]],
    "append": [["-x1000", "-cSN", "-creadType", "T", "h.fasta"], ["-x1000", "-cSN", "T", "g.fasta"]],
}


def header_track_digests(tools, base):
    out = {}
    for tag, cmds in HEADER_TRACK_RUNS.items():
        d = os.path.join(base, tag)
        os.makedirs(d)
        fasta_header_inputs(d)
        for cmd in cmds:
            subprocess.run([os.path.join(tools, "FA2db")] + cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f, m in db_digest(d).items():
            out["%s/%s" % (tag, f)] = m
    return out


def test_fa2db_header_tracks_and_full_hq_filter_equal_reference(built, tmp_path):
    """FA2db -c (header arguments -> tracks) and -Q (FullHqRead only), the part of SURVEY 8(f)3 rounds 1-3 left out: every
    database file and every track against the reference's (tests/golden/fa2db_tracks_ref_md5.txt, made with oracle/_ref/FA2db
    by `make_golden.py fa2db`; compared live as well where oracle/_ref exists)."""
    got = header_track_digests(os.path.join(ROOT, "damar_amd", "bin"), str(tmp_path / "own"))
    want = dict(ln.split()[::-1] for ln in open(os.path.join(GOLDEN, "fa2db_tracks_ref_md5.txt")))
    assert got == want
    assert any(k.endswith(".T.chemistry.data") for k in got) and any(k.startswith("Q/") and k.endswith(".T.readType.anno") for k in got)
    ref = os.path.join(ROOT, "oracle", "_ref")
    if os.path.exists(os.path.join(ref, "FA2db")):
        assert header_track_digests(ref, str(tmp_path / "ref")) == got
    # a malformed argument name stops the tool as it stops the reference
    bad = str(tmp_path / "bad")
    os.makedirs(bad)
    open(os.path.join(bad, "b.fasta"), "w").write(">x bad-name=3\nacgt\n")
    r = subprocess.run([os.path.join(ROOT, "damar_amd", "bin", "FA2db"), "-x1", "-cSN", "U", "b.fasta"], cwd=bad, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "malformed track name" in r.stderr


def db_digest(d, root="T"):
    """md5 of every file of a database; of the .idx only the fields the reference defines (it leaves the
    padding of its records and the tail of the header uninitialised)."""
    import struct
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".fasta") or f.endswith(".fa"):
            continue
        raw = open(os.path.join(d, f), "rb").read()
        if f == ".%s.idx" % root:
            n = (len(raw) - 88) // 32
            raw = raw[:32] + b"".join(struct.pack("<iqqi", *struct.unpack_from("<i4xqqi4x", raw, 88 + 32 * i)) for i in range(n))
        out[f] = hashlib.md5(raw).hexdigest()
    return out


def append_sequence(tools, d):
    """FA2db pb; DBsplit -s1; FA2db plain (the partition is extended); FA2db -a mixed (a new block)."""
    for cmd in (["FA2db", "-x1000", "T", "pb.fasta"], ["DBsplit", "-s1", "T"], ["FA2db", "-x1000", "T", "plain.fa"],
                ["FA2db", "-x1000", "-a", "T", "mixed.fasta"]):
        subprocess.run([os.path.join(tools, cmd[0])] + cmd[1:], cwd=d, check=True, stderr=subprocess.DEVNULL)
    append_sequence.__wrapped_last__ = db_digest(d)
    return append_sequence.__wrapped_last__


def test_fa2db_and_dbsplit_equal_reference(built, tmp_path):
    """bin/FA2db + bin/DBsplit against the reference's tools on the same FASTA files: stub, .bps,
    the defined fields of the .idx, and the seqID / pacbio tracks (tests/golden/fa2db_ref_md5.txt, made with
    oracle/_ref/FA2db and DBsplit by `make_golden.py fa2db`; compared live as well where oracle/_ref exists)."""
    own = str(tmp_path / "own")
    os.makedirs(own)
    files = fasta_inputs(own)
    tools = os.path.join(ROOT, "damar_amd", "bin")
    subprocess.run([os.path.join(tools, "FA2db"), "-x1000", "T"] + files, cwd=own, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(tools, "DBsplit"), "-s1", "T"], cwd=own, check=True)
    got = db_digest(own)
    want = dict(ln.split()[::-1] for ln in open(os.path.join(GOLDEN, "fa2db_ref_md5.txt")))
    assert got == want
    ref = os.path.join(ROOT, "oracle", "_ref")
    if os.path.exists(os.path.join(ref, "FA2db")):
        live = str(tmp_path / "ref")
        os.makedirs(live)
        fasta_inputs(live)
        subprocess.run([os.path.join(ref, "FA2db"), "-x1000", "T"] + files, cwd=live, check=True, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(ref, "DBsplit"), "-s1", "T"], cwd=live, check=True)
        assert db_digest(live) == got
    # the database is usable: the block loader reads it
    from damar_amd import api
    db = api.read_block(os.path.join(own, "T.1"))
    assert db.nreads > 10 and db.maxlen == 12000
    api.lib().damar_close_block(C.byref(db))
    # -f: the same files named in a list file give the same database
    lst = str(tmp_path / "lst")
    os.makedirs(lst)
    fasta_inputs(lst)
    open(os.path.join(lst, "files.txt"), "w").write("\n".join(files) + "\n")
    subprocess.run([os.path.join(tools, "FA2db"), "-x1000", "-ffiles.txt", "T"], cwd=lst, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(tools, "DBsplit"), "-s1", "T"], cwd=lst, check=True)
    d = db_digest(lst)
    d.pop("files.txt")
    assert d == got
    # adding files to an existing, partitioned database (FA2db.c:533-590, 908-975), then -a (new block)
    app = str(tmp_path / "app")
    os.makedirs(app)
    fasta_inputs(app)
    assert append_sequence(tools, app) == dict(ln.split()[::-1] for ln in open(os.path.join(GOLDEN, "fa2db_append_ref_md5.txt")))
    if os.path.exists(os.path.join(ref, "FA2db")):
        live = str(tmp_path / "appref")
        os.makedirs(live)
        fasta_inputs(live)
        assert append_sequence(ref, live) == append_sequence.__wrapped_last__
    # a file cannot be added twice (below)
    # -b: only the longest read of a well (FA2db.c:858-893, the per-record file indices of the seqID track included)
    bst = str(tmp_path / "best")
    os.makedirs(bst)
    fasta_inputs(bst)
    subprocess.run([os.path.join(tools, "FA2db"), "-x1000", "-b", "T"] + files, cwd=bst, check=True, stderr=subprocess.DEVNULL)
    assert db_digest(bst) == dict(ln.split()[::-1] for ln in open(os.path.join(GOLDEN, "fa2db_best_ref_md5.txt")))
    r = subprocess.run([os.path.join(tools, "FA2db"), "T", files[0]], cwd=own, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "already in database" in r.stderr


def test_lamerge_equals_reference(built, tmp_path):
    """LAmerge (the step after daligner in every plan, HPCdaligner.c:790-808): the merged block
    file equals what the reference's utils/LAmerge wrote for the same directory
    (tests/golden/lamerge_ref_md5.txt, generated with oracle/_ref/LAmerge), with and without -s."""
    import hashlib
    import shutil
    import subprocess
    from conftest import GOLDEN, ROOT, link_db, read_case
    exe = os.path.join(ROOT, "damar_amd", "bin", "LAmerge")
    n = 0
    for ln in open(os.path.join(GOLDEN, "lamerge_ref_md5.txt")):
        md5, name, sub, flag = ln.split()
        case = read_case(name)
        w = os.path.join(str(tmp_path), "%s_%s_%d" % (name, sub, n))
        link_db(case["dbdir"], w)
        shutil.copytree(os.path.join(case["lasdir"], sub), os.path.join(w, sub))
        subprocess.run([exe] + (["-s"] if flag == "-s" else []) + ["-n", "8", "G", "m.las", sub], cwd=w, check=True,
                       stdout=subprocess.DEVNULL)
        assert hashlib.md5(open(os.path.join(w, "m.las"), "rb").read()).hexdigest() == md5, ln
        n += 1
    assert n >= 8


@pytest.mark.parametrize("nb,ngpu", [(17, 3), (4, 8), (24, 8), (255, 8), (2, 5)])
def test_node_scheduler_work_list_covers_every_block_pair_once(built, tmp_path, nb, ngpu):
    """`daligner -P plan -G<n>` (host/daligner.c node_main; dalign/daligner.c:958 under HPCdaligner.c:628-788): the work list the
    parent builds before it forks -- every pair of the plan exactly once (split pairs: every part once), regions of about
    equal cost, one region when pairs are split.  DAMAR_NODE_DRYRUN prints it instead of forking (no GPU needed)."""
    import subprocess
    from damar_amd import api
    work = str(tmp_path)
    with open(os.path.join(work, "plan.txt"), "w") as f:
        for a in range(1, nb + 1):
            f.write("daligner -k14 -j16 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))))
    r = subprocess.run([api.daligner_binary(), "-P", "plan.txt", "-G%d" % ngpu], cwd=work, env=dict(os.environ, DAMAR_NODE_DRYRUN="1"),
                       stdout=subprocess.PIPE, text=True, check=True)
    pairs, cost = {}, {}
    for ln in r.stdout.splitlines():
        t = ln.split()
        reg, a = int(t[2]), int(t[4])
        bs = [int(x) for x in t[6:t.index("part")]]
        part, nparts = int(t[t.index("part") + 1]), int(t[t.index("of") + 1])
        assert 1 <= len(bs) <= 8 and all(b <= a for b in bs)
        for b in bs:
            pairs.setdefault((a, b), []).append((part, nparts))
        cost[reg] = cost.get(reg, 0) + int(t[-1])
    assert len(pairs) == nb * (nb + 1) // 2
    for v in pairs.values():
        assert sorted(p for p, _ in v) == list(range(v[0][1]))
    npairs = nb * (nb + 1) // 2
    if npairs >= 2 * ngpu:
        assert len(cost) == ngpu and max(cost.values()) <= 1.35 * min(cost.values())
        if nb >= 200:          # config 4's plan on a node of 8: the regions of the triangle cost the same within 5 %
            assert max(cost.values()) * len(cost) <= 1.05 * sum(cost.values()), cost
    else:
        assert len(cost) == 1 and max(n for v in pairs.values() for _, n in v) > 1

@pytest.mark.parametrize("nb,tile", [(10, 4), (255, 50), (17, 17), (9, 0)])
def test_long_plan_lines_are_dealt_out_by_tiles_without_losing_a_pair(built, tmp_path, nb, tile):
    """`daligner -P plan` on a plan with more blocks than HBM holds indexes for (host/daligner.c tile_plan; the work list of
    HPCdaligner.c:628-788): the lines come out in tiles of T x T block numbers -- every (A, subject) pair of the plan exactly
    once, a line's subject blocks within one tile, tile rows in order and the subject tiles of a row back and forth, so that
    the blocks a stretch of lines names are at most 2 T.  DAMAR_PLAN_DRYRUN prints the lines instead of running them."""
    import subprocess
    from damar_amd import api
    work = str(tmp_path)
    want = set()
    with open(os.path.join(work, "plan.txt"), "w") as f:
        for a in range(1, nb + 1):
            f.write("daligner -k14 -j16 SIM.%d %s\n" % (a, " ".join("SIM.%d" % b for b in range(a, 0, -1))))
            want |= {(a, b) for b in range(1, a + 1)}
    r = subprocess.run([api.daligner_binary(), "-P", "plan.txt"], cwd=work, stdout=subprocess.PIPE, text=True, check=True,
                       env=dict(os.environ, DAMAR_PLAN_DRYRUN="1", DAMAR_PLAN_TILE=str(tile)))
    got, tiles = [], []
    for ln in r.stdout.splitlines():
        t = ln.split()
        assert t[:3] == ["daligner", "-k14", "-j16"]
        a, bs = int(t[3].split(".")[1]), [int(x.split(".")[1]) for x in t[4:]]
        assert bs
        got += [(a, b) for b in bs]
        if tile and tile < nb:
            tj = {(b - 1) // tile for b in bs}
            assert len(tj) == 1
            tiles.append(((a - 1) // tile, tj.pop()))
    assert len(got) == len(want) and set(got) == want
    if tile and tile < nb:
        order = [t for i, t in enumerate(tiles) if i == 0 or tiles[i - 1] != t]
        assert len(order) == len(set(order))                      # a tile is run in one stretch
        assert [t[0] for t in order] == sorted(t[0] for t in order)
        for (i0, j0), (i1, j1) in zip(order, order[1:]):
            assert (i1 == i0 and abs(j1 - j0) == 1) or i1 == i0 + 1
    else:
        assert len(r.stdout.splitlines()) == nb                   # plan order, untouched


def test_teardown_gate_makes_the_next_worker_wait_for_the_one_that_is_leaving(tmp_path):
    """host/damar_gate.h: a worker takes the per-GPU lock before it reports "done" and keeps it until its process is gone;
    a worker that starts waits at the gate (at most two seconds) -- and does not wait for one that is still computing."""
    import time
    src = tmp_path / "gate.c"
    src.write_text('#include "damar_gate.h"\n#include <string.h>\n'
                   'int main(int argc, char **argv)\n'
                   '{ if (strcmp(argv[1], "hold") == 0) { damar_gate_hold(3); usleep(atoi(argv[2]) * 1000); return 0; }\n'
                   '  if (strcmp(argv[1], "busy") == 0) { usleep(atoi(argv[2]) * 1000); return 0; }\n'
                   '  damar_gate_wait(3); return 0; }\n')
    exe = str(tmp_path / "gate")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "damar_amd", "csrc", "host"), "-o", exe, str(src)], check=True)
    env = dict(os.environ, DAMAR_GATE_DIR=str(tmp_path))
    t0 = time.time()
    subprocess.run([exe, "wait"], env=env, check=True)
    assert time.time() - t0 < 0.2                                  # nobody there: the gate is open
    h = subprocess.Popen([exe, "hold", "600"], env=env)
    time.sleep(0.15)
    t0 = time.time()
    subprocess.run([exe, "wait"], env=env, check=True)
    dt = time.time() - t0
    h.wait()
    assert 0.3 < dt < 1.5, dt                                      # waited for the holder's process to go
    b = subprocess.Popen([exe, "busy", "600"], env=env)
    time.sleep(0.1)
    t0 = time.time()
    subprocess.run([exe, "wait"], env=env, check=True)
    assert time.time() - t0 < 0.2                                  # a worker that still computes holds nothing
    b.wait()
    h = subprocess.Popen([exe, "hold", "4000"], env=env)
    time.sleep(0.15)
    t0 = time.time()
    subprocess.run([exe, "wait"], env=env, check=True)
    dt = time.time() - t0
    h.kill()
    h.wait()
    assert 1.8 < dt < 3.0, dt                                      # a holder that does not go away: the gate opens after two seconds
    # a planted symlink under the lock's name is not followed: nothing is created behind it and the gate is simply open
    d2 = tmp_path / "planted"
    d2.mkdir()
    os.symlink(str(d2 / "victim"), str(d2 / "damar_gpu3.teardown"))
    env2 = dict(os.environ, DAMAR_GATE_DIR=str(d2))
    h = subprocess.Popen([exe, "hold", "400"], env=env2)
    time.sleep(0.1)
    t0 = time.time()
    subprocess.run([exe, "wait"], env=env2, check=True)
    assert time.time() - t0 < 0.2
    h.wait()
    assert not (d2 / "victim").exists()
    # and the lock file itself is private to the user
    assert (os.stat(str(tmp_path / "damar_gpu3.teardown")).st_mode & 0o077) == 0


def test_block_read_in_one_stretch_equals_read_by_read(built):
    """damar_read_block: the block's stretch of the .bps file read at once and unpacked 32 bases per step (BMI2 where the CPU
    has it) against the read-by-read path with the 4-base table (db/DB.c:1547-1608 Read_All_Sequences), in fresh processes
    (the choice is made per call from the environment): bases with their terminators and read offsets."""
    code = ("import sys, os, hashlib, ctypes as C\n"
            "sys.path.insert(0, %r)\n"
            "from damar_amd import api\n"
            "L = api.lib()\n"
            "for name in sys.argv[1:]:\n"
            "    db = api.HITS_DB()\n"
            "    assert L.damar_read_block(name.encode(), C.byref(db)) == 0\n"
            "    reads = C.cast(db.reads, C.POINTER(api.HITS_READ))\n"
            "    n = reads[db.nreads].boff\n"
            "    raw = C.string_at(C.c_void_p(db.bases - 1), n + 1)\n"
            "    offs = b''.join(int(reads[i].boff).to_bytes(8, 'little') for i in range(db.nreads + 1))\n"
            "    print(name, db.nreads, hashlib.md5(raw).hexdigest(), hashlib.md5(offs).hexdigest())\n") % ROOT
    names = [os.path.join(ROOT, "tests", "golden", d, "G.1") for d in ("mask_dust", "long", "tandem")] + \
            [os.path.join(ROOT, "tests", "golden", "mask_dust", "G.2")]
    outs = []
    for env in (dict(os.environ), dict(os.environ, DAMAR_DB_READ_BY_READ="1"), dict(os.environ, DAMAR_DB_NO_BMI2="1")):
        r = subprocess.run([os.sys.executable, "-c", code] + names, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and outs[0].count("\n") == len(names), outs


def test_packed_block_unpacks_to_the_unpacked_block_and_its_complement(built):
    """damar_read_block_packed keeps a block as its stretch of the .bps file (the GPU unpacks and reverse-complements it,
    damar_block_upload_packed); damar_unpack_read is the host's view of one read of it (used by the tail where two local
    alignments are bridged): every read of three blocks, both strands, against damar_read_block / damar_complement_copy,
    and the read offsets, frequencies and mirrored mask tracks of the packed block's complement against the unpacked one's."""
    import ctypes as C
    from damar_amd import api
    L = api.lib()

    class Packed(C.Structure):
        _fields_ = [("raw", C.c_void_p), ("nraw", C.c_int64), ("foff", C.POINTER(C.c_uint32)), ("serial", C.c_int64)]
    L.damar_read_block_packed.argtypes = [C.c_char_p, C.POINTER(api.HITS_DB), C.POINTER(Packed)]
    L.damar_read_block_packed.restype = C.c_int
    L.damar_unpack_read.argtypes = [C.POINTER(Packed), C.POINTER(api.HITS_DB), C.c_int, C.c_int, C.c_void_p]
    L.damar_unpack_read.restype = None
    L.damar_free_packed.argtypes = [C.POINTER(Packed)]
    for d, blk in (("mask_dust", "G.1"), ("long", "G.1"), ("tandem", "G.1")):
        name = os.path.join(ROOT, "tests", "golden", d, blk)
        full, cfull = api.HITS_DB(), api.HITS_DB()
        assert L.damar_read_block(name.encode(), C.byref(full)) == 0
        L.damar_complement_copy(C.byref(full), C.byref(cfull))
        pdb, cpdb, pk = api.HITS_DB(), api.HITS_DB(), Packed()
        assert L.damar_read_block_packed(name.encode(), C.byref(pdb), C.byref(pk)) == 0
        assert not pdb.bases and pk.raw and pk.nraw > 0
        L.damar_complement_copy(C.byref(pdb), C.byref(cpdb))
        assert not cpdb.bases
        assert [cpdb.freq[i] for i in range(4)] == [cfull.freq[i] for i in range(4)]
        fr = C.cast(full.reads, C.POINTER(api.HITS_READ))
        pr = C.cast(pdb.reads, C.POINTER(api.HITS_READ))
        assert pdb.nreads == full.nreads and pdb.maxlen == full.maxlen and pdb.totlen == full.totlen
        buf = C.create_string_buffer(full.maxlen + 2)
        for r in range(full.nreads + 1):
            assert pr[r].boff == fr[r].boff
        for r in list(range(0, full.nreads, max(1, full.nreads // 40))) + [full.nreads - 1]:
            n = fr[r].rlen
            assert pr[r].rlen == n
            for comp, src in ((0, full), (1, cfull)):
                L.damar_unpack_read(C.byref(pk), C.byref(pdb), r, comp, C.addressof(buf) + 1)
                want = C.string_at(C.c_void_p(src.bases + fr[r].boff - 1), n + 2)
                assert buf.raw[:n + 2] == want, (d, r, comp)
        L.damar_free_packed(C.byref(pk))
