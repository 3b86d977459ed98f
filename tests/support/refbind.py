"""ctypes binding to the REAL reference compiled in place (oracle/_ref/libimref.so).

Test infrastructure only.  Calls the reference's public attempt_pe_alignment
(src/alignment.h:21-25) with hand-built readaln/readseg structs
(src/readaln.h:13-32) and walks the evidence list it returns
(src/evidence.h:20-36).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libimref.so")
BIN = os.path.join(ROOT, "oracle", "_ref", "indelminer")


def available():
    return os.path.exists(LIB)


class ReadSeg(C.Structure):
    pass


ReadSeg._fields_ = [("next", C.POINTER(ReadSeg)), ("sequence", C.c_void_p),
                    ("oplen", C.c_uint32, 28), ("op", C.c_uint32, 4),
                    ("start", C.c_int32), ("end", C.c_int32)]


class ReadAln(C.Structure):
    _fields_ = [("qname", C.c_void_p), ("tid", C.c_int32), ("strand", C.c_char),
                ("index", C.c_char), ("qual", C.c_uint8), ("segments", C.POINTER(ReadSeg))]


class Evidence(C.Structure):
    pass


Evidence._fields_ = [("next", C.POINTER(Evidence)), ("type", C.c_int), ("variantclass", C.c_int),
                     ("strand", C.c_char), ("qual", C.c_uint8), ("qname", C.c_void_p),
                     ("aln1", C.POINTER(ReadSeg)), ("aln2", C.POINTER(ReadSeg)), ("aln3", C.POINTER(ReadSeg)),
                     ("b1", C.c_int32), ("b2", C.c_int32), ("mindelsize", C.c_int32), ("max", C.c_int32),
                     ("isused", C.c_int)]


class Ref:
    def __init__(self):
        self.lib = C.CDLL(LIB)
        L = self.lib
        L.ckallocz.restype = C.c_void_p
        L.ckallocz.argtypes = [C.c_size_t]
        L.ckfree.argtypes = [C.c_void_p]
        L.attempt_pe_alignment.restype = C.POINTER(Evidence)
        L.attempt_pe_alignment.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32,
                                           C.POINTER(C.c_int), C.POINTER(ReadAln)]
        self.set_params()
        # debug_file must be a valid FILE* even with debug_flag off
        libc = C.CDLL(None)
        C.c_int.in_dll(L, "debug_flag").value = 0

    def set_params(self, klength=6, numgaps=0, maxdelsize=1000, ethreshold=10):
        L = self.lib
        C.c_uint.in_dll(L, "klength").value = klength
        C.c_uint.in_dll(L, "numgaps").value = numgaps
        C.c_uint.in_dll(L, "maxdelsize").value = maxdelsize
        C.c_uint.in_dll(L, "ethreshold").value = ethreshold
        C.c_uint32.in_dll(L, "seed_mask").value = (1 << (2 * (klength - 1))) - 1  # src/indelminer.c:1071

    def _cstr(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        p = self.lib.ckallocz(len(b) + 1)
        C.memmove(p, b, len(b))
        return p

    @staticmethod
    def _segs(p):
        out = []
        while p:
            s = p.contents
            seq = C.string_at(s.sequence).decode() if s.sequence else None
            out.append((int(s.op), int(s.oplen), int(s.start), int(s.end), seq))
            p = s.next
        return out

    def realign(self, contig_buf, anchor, range_max, read, qual=60, strand=b"+"):
        """contig_buf: ctypes char buffer (NUL terminated).  Returns None (reference
        returned NULL) or a list of evidence dicts in the order of the returned list."""
        L = self.lib
        seg = C.cast(L.ckallocz(C.sizeof(ReadSeg)), C.POINTER(ReadSeg))
        seg.contents.sequence = self._cstr(read)
        seg.contents.oplen = len(read)
        seg.contents.op = 4
        seg.contents.start = -1
        seg.contents.end = -1
        rln = ReadAln()
        rln.qname = self._cstr("q")
        rln.tid = -1
        rln.strand = strand
        rln.index = b"1"
        rln.qual = qual
        rln.segments = seg
        seqs = (C.c_char_p * 1)(C.cast(contig_buf, C.c_char_p))
        rng = (C.c_int * 2)(0, range_max)
        ev = L.attempt_pe_alignment(seqs, 0, anchor, rng, C.byref(rln))
        out = None
        if ev:
            out = []
            p = ev
            while p:
                e = p.contents
                out.append(dict(cls=int(e.variantclass), type=int(e.type), b1=int(e.b1), b2=int(e.b2),
                                qual=int(e.qual), strand=e.strand,
                                aln1=self._segs(e.aln1), aln2=self._segs(e.aln2), aln3=self._segs(e.aln3)))
                p = e.next
        L.ckfree(rln.qname)
        return out
