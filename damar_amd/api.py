"""ctypes mirror of the overlap path's C interface.

Same names and argument meaning as the reference's ``dalign/filter.h:64-70``,
``dalign/align.h:223-259, 416-432`` and the subset of ``db/DB.h`` that
``dalign/daligner.c`` uses; see include/damar_*.h for the citations.  Every call goes
straight into libdamar_hip.so -- nothing is computed in Python.
"""
import ctypes as C
import os

from .lib import load, bin_path

c_int64 = C.c_int64


class HITS_READ(C.Structure):          # db/DB.h:319-326
    _fields_ = [("rlen", C.c_int), ("boff", c_int64), ("coff", c_int64), ("flags", C.c_int)]


class HITS_DB(C.Structure):            # db/DB.h:361-389
    _fields_ = [("ureads", C.c_int), ("freq", C.c_float * 4), ("maxlen", C.c_int),
                ("totlen", c_int64), ("nreads", C.c_int), ("part", C.c_int), ("ufirst", C.c_int),
                ("path", C.c_char_p), ("loaded", C.c_int), ("bases", C.c_void_p),
                ("reads", C.POINTER(HITS_READ)), ("tracks", C.c_void_p)]


class SimParams(C.Structure):          # include/damar_db.h damar_sim_params
    _fields_ = [("genome_mbp", C.c_double), ("coverage", C.c_double), ("bias", C.c_double),
                ("seed", C.c_int), ("rmean", C.c_int), ("rsdev", C.c_int), ("rshort", C.c_int),
                ("erate", C.c_double), ("block_mbp", C.c_int), ("min_len", C.c_int),
                ("tandem_frac", C.c_double), ("max_blocks", C.c_int)]


T_NAMES = ["tuples", "ksort", "table", "merge", "ssort", "work", "report", "d2h", "tail"]

class MatchJob(C.Structure):           # include/damar_hip.h damar_match_job
    _fields_ = [("ablock", C.POINTER(HITS_DB)), ("bblock", C.POINTER(HITS_DB)),
                ("aidx", C.c_void_p), ("bidx", C.c_void_p),
                ("self_", C.c_int), ("comp", C.c_int),
                ("spec", C.c_void_p), ("counts", c_int64 * 3)]


_proto_done = False


def _lib():
    global _proto_done
    L = load()
    if not _proto_done:
        L.damar_read_block.argtypes = [C.c_char_p, C.POINTER(HITS_DB)]
        L.damar_read_block.restype = C.c_int
        L.damar_close_block.argtypes = [C.POINTER(HITS_DB)]
        L.damar_complement_block.argtypes = [C.POINTER(HITS_DB), C.c_int]
        L.damar_complement_block.restype = C.POINTER(HITS_DB)
        L.damar_sim_defaults.argtypes = [C.POINTER(SimParams)]
        L.damar_sim_write_db.argtypes = [C.POINTER(SimParams), C.c_char_p, C.c_char_p]
        L.damar_sim_write_db.restype = C.c_int
        L.damar_get_dir.argtypes = [C.c_int, C.c_int]
        L.damar_get_dir.restype = C.c_void_p
        L.Set_Filter_Params.argtypes = [C.c_int] * 5
        L.Set_Filter_Params.restype = C.c_int
        L.Sort_Kmers.argtypes = [C.POINTER(HITS_DB), C.POINTER(C.c_int)]
        L.Sort_Kmers.restype = C.c_void_p
        L.Match_Filter.argtypes = [C.c_char_p, C.POINTER(HITS_DB), C.c_char_p, C.POINTER(HITS_DB),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.New_Align_Spec.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int]
        L.New_Align_Spec.restype = C.c_void_p
        L.Free_Align_Spec.argtypes = [C.c_void_p]
        L.Write_Overlap_Buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.Reset_Overlap_Buffer.argtypes = [C.c_void_p]
        L.damar_hip_init.argtypes = [C.c_int]
        L.damar_hip_init.restype = C.c_int
        L.damar_hip_device_name.restype = C.c_char_p
        L.damar_block_upload.argtypes = [C.POINTER(HITS_DB)]
        L.damar_block_upload.restype = C.c_void_p
        L.damar_block_free.argtypes = [C.c_void_p]
        L.damar_load_masks.argtypes = [C.POINTER(HITS_DB), C.POINTER(C.c_char_p), C.c_int]
        L.damar_load_masks.restype = C.c_int
        L.damar_async_d2h_ms.restype = C.c_double
        L.damar_index_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.damar_index_build.restype = C.c_void_p
        L.damar_index_free.argtypes = [C.c_void_p]
        L.damar_index_bytes.argtypes = [C.c_void_p]
        L.damar_index_bytes.restype = C.c_uint64
        L.damar_complement_copy.argtypes = [C.POINTER(HITS_DB), C.POINTER(HITS_DB)]
        L.damar_free_complement.argtypes = [C.POINTER(HITS_DB)]
        L.damar_async_drain.argtypes = []
        L.damar_set_bread_range.argtypes = [C.c_int, C.c_int]
        L.damar_index_download.argtypes = [C.c_void_p, C.c_void_p]
        L.damar_match.argtypes = [C.POINTER(HITS_DB), C.POINTER(HITS_DB), C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p, C.POINTER(c_int64)]
        L.damar_match_batch.argtypes = [C.POINTER(MatchJob), C.c_int]
        L.damar_set_async.argtypes = [C.c_int]
        L.damar_async_totals.argtypes = [C.POINTER(c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.damar_async_counts.argtypes = [C.POINTER(c_int64), C.POINTER(C.c_double), C.POINTER(c_int64)]
        L.damar_wave_totals.argtypes = [C.POINTER(c_int64)] * 3
        L.damar_write_overlaps.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.damar_tandem_set_params.argtypes = [C.c_int] * 4
        L.damar_tandem_set_params.restype = C.c_int
        L.Match_Self.argtypes = [C.c_char_p, C.POINTER(HITS_DB), C.c_void_p]
        L.damar_match_self.argtypes = [C.POINTER(HITS_DB), C.c_void_p, C.c_void_p, C.POINTER(c_int64)]
        L.damar_last_seeds.argtypes = [C.c_void_p, c_int64]
        L.damar_last_seeds.restype = c_int64
        L.damar_local_alignment_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                                  C.POINTER(c_int64), C.POINTER(C.c_uint16), c_int64]
        L.damar_local_alignment_batch.restype = C.c_int
        L.damar_last_timings.argtypes = [C.POINTER(C.c_double)]
        L.damar_last_counters.argtypes = [C.POINTER(c_int64)]
        L.damar_bench_sort_u32.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint32]
        L.damar_bench_sort_u32.restype = C.c_double
        _proto_done = True
    return L


def set_globals(verbose=0, minover=2000, symmetric=1, identity=0, hgap_min=0, biased=0):
    """The globals daligner.c:131-140 shares with filter.c (MINOVER is already doubled)."""
    L = _lib()
    C.c_int.in_dll(L, "VERBOSE").value = verbose
    C.c_int.in_dll(L, "MINOVER").value = minover
    C.c_int.in_dll(L, "SYMMETRIC").value = symmetric
    C.c_int.in_dll(L, "IDENTITY").value = identity
    C.c_int.in_dll(L, "HGAP_MIN").value = hgap_min
    C.c_int.in_dll(L, "BIASED").value = biased


def sim_write_db(directory, root, genome_mbp, coverage=20., seed=1, erate=.15, block_mbp=200,
                 rmean=10000, rsdev=2000, rshort=4000, tandem_frac=0., max_blocks=0):
    """db/simulator.c | FA2db | DBsplit equivalent (include/damar_db.h); returns #blocks."""
    L = _lib()
    p = SimParams()
    L.damar_sim_defaults(C.byref(p))
    p.genome_mbp, p.coverage, p.seed, p.erate, p.block_mbp = genome_mbp, coverage, seed, erate, block_mbp
    p.rmean, p.rsdev, p.rshort = rmean, rsdev, rshort
    p.tandem_frac = tandem_frac
    p.max_blocks = max_blocks
    nb = L.damar_sim_write_db(C.byref(p), directory.encode(), root.encode())
    if nb < 0:
        raise RuntimeError("damar_sim_write_db failed")
    return nb


def read_block(name):
    L = _lib()
    db = HITS_DB()
    if L.damar_read_block(name.encode(), C.byref(db)) != 0:
        raise RuntimeError("cannot read block %s" % name)
    return db


def get_dir(run, block):
    L = _lib()
    p = L.damar_get_dir(run, block)
    return C.string_at(p).decode()


def timings():
    L = _lib()
    a = (C.c_double * len(T_NAMES))()
    L.damar_last_timings(a)
    return dict(zip(T_NAMES, list(a)))


def counters():
    L = _lib()
    a = (c_int64 * 8)()
    L.damar_last_counters(a)
    return list(a)


def daligner_binary():
    p = bin_path("daligner")
    if not os.path.exists(p):
        raise RuntimeError("%s not built" % p)
    return p


def lib():
    return _lib()
