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
        srcs = [os.path.join(ODIR, f) for f in ("im_oracle.c", "im_oracle_triage.c", "im_oracle.h")]
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
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


class Triage(C.Structure):
    _fields_ = [("cls", C.c_int32), ("revcomp", C.c_int32), ("range_max", C.c_int32), ("qual", C.c_int32),
                ("strand", C.c_int32), ("tid", C.c_int32), ("anchor", C.c_int32), ("l_seq", C.c_int32),
                ("n_ev", C.c_int32), ("ev_cls", C.c_int32 * MAX_EV), ("ev_b1", C.c_int32 * MAX_EV), ("ev_b2", C.c_int32 * MAX_EV), ("want", C.c_int32)]


def triage_records(raw, rec_off, rg_names, rg_range_max, qthreshold=10, eth_vcf=10, maxpedelsize=1000000):
    """imo_triage_record over every record of a device-layout buffer.  Returns a list of
    (Triage, bases bytes or None)."""
    import numpy as np
    L = lib()
    L.imo_triage_record.restype = None
    L.imo_triage_record.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p,
                                    C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(Triage), C.c_char_p]
    n = len(rec_off) - 1
    names = (C.c_char_p * max(len(rg_names), 1))(*[x.encode() for x in rg_names])
    rm = np.ascontiguousarray(rg_range_max, dtype=np.int32)
    raw = np.ascontiguousarray(raw)
    base = raw.ctypes.data
    out = []
    for i in range(n):
        t = Triage()
        ln = int(rec_off[i + 1]) - int(rec_off[i])
        buf = C.create_string_buffer(4096)
        L.imo_triage_record(base + int(rec_off[i]), ln, len(rg_names), names, rm.ctypes.data, qthreshold, eth_vcf, maxpedelsize,
                            C.byref(t), buf)
        out.append((t, buf.raw[:t.l_seq] if t.cls in (2, 3) else None))
    return out


def depth_of(raw, rec_off, tid, clen):
    import numpy as np
    L = lib()
    L.imo_depth_add.restype = None
    L.imo_depth_add.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_int64]
    raw = np.ascontiguousarray(raw)
    depth = np.zeros(clen, dtype=np.int32)
    for i in range(len(rec_off) - 1):
        L.imo_depth_add(raw.ctypes.data + int(rec_off[i]), int(rec_off[i + 1]) - int(rec_off[i]), tid, depth.ctypes.data, clen)
    return depth


def flush_cut(cls, b1, b2, consumed, marker, flush_id):
    import numpy as np
    L = lib()
    L.imo_flush_cut.restype = C.c_int32
    L.imo_flush_cut.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    cls = np.ascontiguousarray(cls, np.int32); b1 = np.ascontiguousarray(b1, np.int32); b2 = np.ascontiguousarray(b2, np.int32)
    assert consumed.dtype == np.int32 and consumed.flags.c_contiguous
    return L.imo_flush_cut(len(cls), cls.ctypes.data, b1.ctypes.data, b2.ctypes.data, consumed.ctypes.data, marker, flush_id)


def flush_nohistory(marker, ids, last, cls, b1, b2, arr):
    """imo_flush_nohistory: consumed[] for the whole flush list at once, or None when a contig's markers decrease"""
    import numpy as np
    L = lib()
    L.imo_flush_nohistory.restype = C.c_int32
    L.imo_flush_nohistory.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] + [C.c_void_p] * 5
    a = [np.ascontiguousarray(x, np.int32) for x in (marker, ids, last, cls, b1, b2, arr)]
    consumed = np.full(len(a[3]), -7, np.int32)
    rc = L.imo_flush_nohistory(len(a[0]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[3]),
                               a[3].ctypes.data, a[4].ctypes.data, a[5].ctypes.data, a[6].ctypes.data, consumed.ctypes.data)
    return consumed if rc == 0 else None
