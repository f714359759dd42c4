"""The C-ABI library loads on a CPU-only machine and exports every function that
include/*.h declares (no compute is called here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT

DECL = re.compile(r"^\s*(?:const\s+)?(?:unsigned\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*+\s*|\s+)\**\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", re.M)


def declared_functions():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"#define[^\n]*\n", "\n", txt)
        for m in DECL.finditer(txt):
            n = m.group(1)
            if n not in ("defined", "sizeof", "COMP", "ACOMP"):
                names.add(n)
    return names


def test_library_exports_every_declared_symbol(built):
    lib = os.path.join(ROOT, "damar_amd", "libdamar_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    decl = declared_functions()
    assert len(decl) > 30
    missing = sorted(n for n in decl if n not in exported)
    assert missing == []


def test_reference_interface_names_and_globals(built):
    L = ctypes.CDLL(os.path.join(ROOT, "damar_amd", "libdamar_hip.so"))
    for fn in ("Set_Filter_Params", "Sort_Kmers", "Match_Filter", "New_Align_Spec", "Write_Overlap_Buffer",
               "Reset_Overlap_Buffer", "AddOverlapToBuffer", "OVL_IO_Buffer", "Compress_TraceTo8",
               "New_Work_Data", "Free_Work_Data", "Local_Alignment", "Match_Self"):
        assert hasattr(L, fn)
    for g in ("BIASED", "VERBOSE", "MINOVER", "HGAP_MIN", "SYMMETRIC", "IDENTITY"):
        ctypes.c_int.in_dll(L, g)
    for g in ("MEM_LIMIT", "MEM_PHYSICAL"):
        ctypes.c_uint64.in_dll(L, g)
    L.Set_Filter_Params.restype = ctypes.c_int
    assert L.Set_Filter_Params(1, 6, 0, 35, 4) == 1          # filter.c:173-174 illegal k
    assert L.Set_Filter_Params(14, 6, 0, 35, 4) == 0


def test_struct_layouts_match_reference():
    from damar_amd import api
    assert ctypes.sizeof(api.HITS_DB) == 88
    assert ctypes.sizeof(api.HITS_READ) == 32
    assert api.HITS_DB.maxlen.offset == 20 and api.HITS_DB.totlen.offset == 24
    assert api.HITS_DB.nreads.offset == 32 and api.HITS_DB.bases.offset == 64
    assert api.HITS_READ.boff.offset == 8 and api.HITS_READ.flags.offset == 24


def test_product_does_not_link_the_oracle(built):
    lib = os.path.join(ROOT, "damar_amd", "libdamar_hip.so")
    out = subprocess.run(["nm", "-D", lib], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "oracle_" not in out
    ldd = subprocess.run(["ldd", os.path.join(ROOT, "damar_amd", "bin", "daligner")], stdout=subprocess.PIPE, text=True).stdout
    assert "liboracle" not in ldd


def test_tandem_library_is_the_second_boundary(built):
    """scrub/tandem.h:54-60: a scrub/datander.c that links libdamar_tandem.so (before libdamar_hip.so) instead of
    scrub/tandem.c finds the FOUR-argument Set_Filter_Params and SORT_PATH there, Match_Self and everything of align.h in
    libdamar_hip.so, which the tandem library pulls in itself."""
    tan = os.path.join(ROOT, "damar_amd", "libdamar_tandem.so")
    out = subprocess.run(["nm", "-D", "--defined-only", tan], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = {ln.split()[-1]: ln.split()[-2] for ln in out.splitlines() if len(ln.split()) >= 2}
    assert exported.get("Set_Filter_Params") == "T"
    assert exported.get("SORT_PATH") in ("D", "B")
    assert "Match_Self" not in exported                     # one definition, in the main library
    needed = subprocess.run(["readelf", "-d", tan], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "libdamar_hip.so" in needed
    L = ctypes.CDLL(tan)
    L.Set_Filter_Params.restype = ctypes.c_int
    L.Set_Filter_Params.argtypes = [ctypes.c_int] * 4
    assert L.Set_Filter_Params(1, 4, 35, 4) == 1            # tandem.c:160-162: k <= 1 is illegal
    assert L.Set_Filter_Params(12, 4, 35, 4) == 0
    assert ctypes.c_char_p.in_dll(L, "SORT_PATH").value == b"/tmp"
    H = ctypes.CDLL(os.path.join(ROOT, "damar_amd", "libdamar_hip.so"))
    assert hasattr(H, "Match_Self")
