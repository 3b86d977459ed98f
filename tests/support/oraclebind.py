"""ctypes binding to oracle/liboracle.so (our CPU restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "liboracle.so")

MAX_OPS = 256
MAX_EV = 32


class Params(C.Structure):
    _fields_ = [("klength", C.c_uint32), ("numgaps", C.c_uint32),
                ("maxdelsize", C.c_uint32), ("ethreshold", C.c_uint32)]


class BandAln(C.Structure):
    _fields_ = [("r1", C.c_int32), ("r2", C.c_int32), ("q1", C.c_int32), ("q2", C.c_int32),
                ("n_ops", C.c_int32), ("ops", C.c_uint32 * MAX_OPS),
                ("low", C.c_int32), ("up", C.c_int32), ("mismatches", C.c_int32)]


class Ev(C.Structure):
    _fields_ = [("cls", C.c_int32), ("b1", C.c_int32), ("b2", C.c_int32), ("seg", C.c_int32),
                ("read_off", C.c_int32), ("lflank", C.c_int32), ("rflank", C.c_int32),
                ("nd_print", C.c_int32), ("nd_filter", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("ref_start", C.c_int32), ("n_ops", C.c_int32),
                ("ops", C.c_uint32 * MAX_OPS), ("n_ev", C.c_int32), ("ev", Ev * MAX_EV),
                ("n_band", C.c_int32), ("win_bytes", C.c_int32 * 2), ("piece_bytes", C.c_int32 * 2),
                ("piece", BandAln * 2)]


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ODIR, "im_oracle.c")):
            build()
        L = C.CDLL(LIB)
        L.imo_realign.restype = C.c_int
        L.imo_realign.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_char_p, C.c_int32, C.POINTER(Result)]
        L.imo_find_best_band.restype = C.c_int
        L.imo_find_best_band.argtypes = [C.POINTER(Params), C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_char_p, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.imo_cluster_sr.restype = C.c_int32
        L.imo_realign_batch.restype = C.c_int
        _lib = L
    return _lib


def params(klength=6, numgaps=0, maxdelsize=1000, ethreshold=10):
    return Params(klength, numgaps, maxdelsize, ethreshold)


def segments(res):
    """Expand a Result's packed ops into (op, len, start, end, read_off) tuples
    following new_readseg's coordinate rules (src/readaln.c:24-99)."""
    out = []
    ref = res.ref_start
    rd = 0
    for i in range(res.n_ops):
        w = res.ops[i]
        op, ln = w & 15, w >> 4
        start = ref
        if op in (7, 8, 0, 2):
            ref += ln
        out.append((op, ln, start, ref, rd))
        if op != 2:
            rd += ln
    return out


def realign(P, contig_bytes, contig_len, anchor, range_max, read):
    r = Result()
    st = lib().imo_realign(C.byref(P), contig_bytes, contig_len, anchor, range_max,
                           read.encode() if isinstance(read, str) else read, len(read), C.byref(r))
    return st, r
