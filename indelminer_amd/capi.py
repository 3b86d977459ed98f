"""ctypes mirror of include/indelminer_amd.h.

Thin plumbing for the tests and bench.py: the product is the HIP library
(indelminer_amd/libindelminer_amd.so) and the C host driver above it.  There is
no CPU fallback here: if the library is missing or no gfx950 device is present,
everything raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_OPS = 64
MAX_EV = 4
MAX_READ = 1020
SHORT_READ = 255        # reads beyond it take the second realign launch (Context.expect_read_length)

IM_OK = 0
E_ARG, E_NOGPU, E_HIP, E_UNSUPPORTED, E_ABORT, E_OVERFLOW = -1, -2, -3, -4, -5, -6
ST_NONE, ST_EVIDENCE, ST_ABORT, ST_OVERFLOW, ST_UNSUPPORTED = 0, 1, -1, -2, -3


class Params(C.Structure):
    _fields_ = [("klength", C.c_uint32), ("numgaps", C.c_uint32),
                ("maxdelsize", C.c_uint32), ("ethreshold", C.c_uint32)]


class Evidence(C.Structure):
    _fields_ = [("cls", C.c_int32), ("b1", C.c_int32), ("b2", C.c_int32), ("seg", C.c_int32),
                ("read_off", C.c_int32), ("lflank", C.c_int32), ("rflank", C.c_int32),
                ("nd_print", C.c_int32), ("nd_filter", C.c_int32)]


class BandAln(C.Structure):
    _fields_ = [("r1", C.c_int32), ("r2", C.c_int32), ("q1", C.c_int32), ("q2", C.c_int32),
                ("low", C.c_int32), ("votes", C.c_int32), ("win_bytes", C.c_int32), ("piece_bytes", C.c_int32)]


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("ref_start", C.c_int32), ("n_ops", C.c_int32), ("n_ev", C.c_int32),
                ("n_band", C.c_int32), ("reserved", C.c_int32 * 7),
                ("band", BandAln * 2), ("ev", Evidence * MAX_EV), ("ops", C.c_uint32 * MAX_OPS)]


assert C.sizeof(ReadResult) == 512

# numpy view of the same record
RESULT_DTYPE = np.dtype([
    ("status", "<i4"), ("ref_start", "<i4"), ("n_ops", "<i4"), ("n_ev", "<i4"), ("n_band", "<i4"),
    ("reserved", "<i4", (7,)),
    ("band", [("r1", "<i4"), ("r2", "<i4"), ("q1", "<i4"), ("q2", "<i4"), ("low", "<i4"),
              ("votes", "<i4"), ("win_bytes", "<i4"), ("piece_bytes", "<i4")], (2,)),
    ("ev", [("cls", "<i4"), ("b1", "<i4"), ("b2", "<i4"), ("seg", "<i4"), ("read_off", "<i4"),
            ("lflank", "<i4"), ("rflank", "<i4"), ("nd_print", "<i4"), ("nd_filter", "<i4")], (MAX_EV,)),
    ("ops", "<u4", (MAX_OPS,)),
])
assert RESULT_DTYPE.itemsize == 512


class ReadBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("bases", C.c_void_p), ("base_off", C.c_void_p),
                ("tid", C.c_void_p), ("anchor", C.c_void_p), ("range_max", C.c_void_p)]


class DevBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("bases", C.c_void_p), ("base_off", C.c_void_p), ("read_len", C.c_void_p),
                ("tid", C.c_void_p), ("anchor", C.c_void_p), ("range_max", C.c_void_p), ("out", C.c_void_p),
                ("ev_cls", C.c_void_p), ("ev_b1", C.c_void_p), ("ev_b2", C.c_void_p)]


class DevRecords(C.Structure):
    _fields_ = [("n", C.c_int32), ("raw", C.c_void_p), ("rec_off", C.c_void_p), ("rec_base", C.c_int32)]


class TriageParams(C.Structure):
    _fields_ = [("qthreshold", C.c_int32), ("ethreshold_vcfcheck", C.c_uint32), ("maxpedelsize", C.c_uint32),
                ("want_depth", C.c_int32), ("defer_ranges", C.c_int32), ("restart", C.c_int32)]


class DevCands(C.Structure):
    _fields_ = [("batch", DevBatch), ("cand_rec", C.c_void_p), ("counters", C.c_void_p), ("rec_class", C.c_void_p),
                ("cap_cand", C.c_int32), ("cap_bases", C.c_int64), ("consumed", C.c_void_p)]


REC_SKIP, REC_COUNTED, REC_CAND_UNMAPPED, REC_CAND_PROPER, REC_PE = 0, 1, 2, 3, 4


class IMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("indelminer_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def library_path():
    return _build.LIB


def lib():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = os.environ.get("INDELMINER_AMD_LIB", _build.LIB)      # diagnostic builds (profiles/) only
        if not os.path.exists(path):
            raise IMError(E_NOGPU, "libindelminer_amd.so is not built; run __graft_entry__.build()")
        L = C.CDLL(path)
        L.im_last_error.restype = C.c_char_p
        L.im_last_error.argtypes = [C.c_void_p]
        L.im_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.im_ctx_destroy.argtypes = [C.c_void_p]
        L.im_ctx_destroy.restype = None
        L.im_set_reference.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
        L.im_realign_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(ReadBatch), C.c_void_p]
        L.im_cluster_sr.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.im_dev_realign.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(DevBatch), C.c_void_p]
        L.im_expect_read_length.argtypes = [C.c_void_p, C.c_int32]
        L.im_dev_compact_results.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_dev_cluster_scratch_bytes.restype = C.c_size_t
        L.im_dev_cluster_scratch_bytes.argtypes = [C.c_int32]
        L.im_dev_gather_scratch_bytes.restype = C.c_size_t
        L.im_dev_gather_scratch_bytes.argtypes = [C.c_int32]
        L.im_dev_cluster_sr.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_gather_evidence.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_cluster_slots.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_dev_cluster_records.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]
        L.im_comm_unique_id.argtypes = [C.c_void_p]
        L.im_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.im_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_comm_destroy.argtypes = [C.c_void_p]
        L.im_comm_destroy.restype = None
        L.im_comm_last_error.restype = C.c_char_p
        L.im_dev_cluster_hist_scratch_bytes.restype = C.c_size_t
        L.im_dev_cluster_hist_scratch_bytes.argtypes = [C.c_int32]
        L.im_dev_cluster_hist_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_cluster_hist.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_depth_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.im_depth_query.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_support_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.im_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        L.im_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.im_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.im_ctx_stream.restype = C.c_void_p
        L.im_ctx_stream.argtypes = [C.c_void_p]
        L.im_stream_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.im_timer_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.im_timer_destroy.argtypes = [C.c_void_p]
        L.im_timer_destroy.restype = None
        L.im_timer_start.argtypes = [C.c_void_p, C.c_void_p]
        L.im_timer_stop.argtypes = [C.c_void_p, C.c_void_p]
        L.im_timer_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.im_stream_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.im_stream_destroy.argtypes = [C.c_void_p, C.c_void_p]
        L.im_event_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.im_event_destroy.argtypes = [C.c_void_p]
        L.im_event_destroy.restype = None
        L.im_event_record.argtypes = [C.c_void_p, C.c_void_p]
        L.im_event_sync.argtypes = [C.c_void_p]
        L.im_stream_follow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_dev_realign_keep.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(DevBatch), C.c_void_p]
        L.im_set_insert_ranges.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p]
        L.im_dev_triage_scratch_bytes.restype = C.c_size_t
        L.im_dev_triage_scratch_bytes.argtypes = [C.c_int32]
        L.im_dev_triage.argtypes = [C.c_void_p, C.POINTER(TriageParams), C.POINTER(DevRecords), C.POINTER(DevCands),
                                    C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_flush_cut.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.im_dev_groupby_scratch_bytes.restype = C.c_size_t
        L.im_dev_groupby_scratch_bytes.argtypes = [C.c_int32]
        L.im_dev_cluster_groupby.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_realign_n.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(DevBatch), C.c_void_p, C.c_int32, C.c_void_p]
        L.im_dev_flush_cut_rec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.im_dev_flush_cuts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.im_dev_flushgroup_scratch_bytes.restype = C.c_size_t
        L.im_dev_flushgroup_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
        L.im_dev_flushgroup_scratch_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_flush_groupby.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_triage_scratch_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_groupby_scratch_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_dev_cluster_groupby_n.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_depth_enable.argtypes = [C.c_void_p]
        L.im_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.im_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.im_dev_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_depth_scan.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.im_depth_reset.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.im_depth_query_tid.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.im_dev_memset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        L.im_dev_copy_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.im_capture_begin.argtypes = [C.c_void_p, C.c_void_p]
        L.im_capture_end.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.im_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.im_graph_destroy.argtypes = [C.c_void_p]
        L.im_graph_destroy.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DevBuf:
    """A device allocation owned through the C ABI."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx._check(lib().im_dev_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(lib().im_dev_upload(self.ctx.h, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(lib().im_dev_download(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().im_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class DevView:
    """A window into somebody else's device allocation (same download interface as DevBuf)."""

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, int(nbytes)

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(lib().im_dev_download(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out


class Event:
    """Stream-to-stream dependency: record on one stream, wait on another."""

    def __init__(self, ctx):
        self.ctx = ctx
        p = C.c_void_p()
        ctx._check(lib().im_event_create(ctx.h, C.byref(p)))
        self.h = p.value

    def record(self, stream):
        self.ctx._check(lib().im_event_record(self.h, stream))

    def wait(self, stream):
        self.ctx._check(lib().im_stream_wait_event(self.ctx.h, stream, self.h))

    def sync(self):
        self.ctx._check(lib().im_event_sync(self.h))

    def follow(self, src, dst):
        """record on stream src, make stream dst wait"""
        self.ctx._check(lib().im_stream_follow(self.h, src, dst))

    def close(self):
        if self.h:
            lib().im_event_destroy(self.h)
            self.h = None


def new_stream(ctx):
    p = C.c_void_p()
    ctx._check(lib().im_stream_create(ctx.h, C.byref(p)))
    return p.value


class Graph:
    """A captured sequence of im_dev_* launches: `with Graph.capture(ctx) as g: ...launches...`, then g.launch()."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = None

    @classmethod
    def capture(cls, ctx):
        return cls(ctx)

    def __enter__(self):
        self.ctx._check(lib().im_capture_begin(self.ctx.h, self.ctx.stream))
        return self

    def __exit__(self, et, ev, tb):
        p = C.c_void_p()
        rc = lib().im_capture_end(self.ctx.h, self.ctx.stream, C.byref(p))
        if et is None:
            self.ctx._check(rc)
            self.h = p.value
        return False

    def launch(self):
        self.ctx._check(lib().im_graph_launch(self.h, self.ctx.stream))

    def close(self):
        if self.h:
            lib().im_graph_destroy(self.h)
            self.h = None


class Timer:
    def __init__(self, ctx):
        self.ctx = ctx
        p = C.c_void_p()
        ctx._check(lib().im_timer_create(ctx.h, C.byref(p)))
        self.h = p.value

    def start(self, stream):
        self.ctx._check(lib().im_timer_start(self.h, stream))

    def stop(self, stream):
        self.ctx._check(lib().im_timer_stop(self.h, stream))

    def elapsed_ms(self):
        ms = C.c_float()
        self.ctx._check(lib().im_timer_elapsed_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.h:
            lib().im_timer_destroy(self.h)
            self.h = None


class Context:
    """One GPU.  Mirrors im_ctx_* / im_set_reference / im_realign_batch / im_cluster_sr."""

    def __init__(self, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.im_ctx_create(device, C.byref(h))
        if rc != IM_OK:
            raise IMError(rc, L.im_last_error(None).decode())
        self.h = h.value
        self.stream = L.im_ctx_stream(self.h)
        self._keep = None

    def _check(self, rc, allow=()):
        if rc != IM_OK and rc not in allow:
            raise IMError(rc, lib().im_last_error(self.h).decode())
        return rc

    def close(self):
        if self.h:
            lib().im_ctx_destroy(self.h)
            self.h = None

    def set_reference(self, contigs):
        """contigs: list of bytes (upper-cased ASCII, as read_reference keeps them)."""
        n = len(contigs)
        arr = (C.c_char_p * n)(*contigs)
        lens = (C.c_int64 * n)(*[len(c) for c in contigs])
        self._check(lib().im_set_reference(self.h, n, arr, lens))
        self.contig_len = [len(c) for c in contigs]

    def expect_read_length(self, max_len):
        """reads of 256 .. MAX_READ bases occur: every realign launch is followed by the long-read kernel"""
        self._check(lib().im_expect_read_length(self.h, int(max_len)))

    def realign_batch(self, params, reads, tid, anchor, range_max, allow=()):
        """reads: list of bytes.  Returns (rc, numpy structured array of RESULT_DTYPE)."""
        n = len(reads)
        bases = np.frombuffer(b"".join(reads) + b"\0" * 8, dtype=np.uint8).copy()
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(r) for r in reads], out=off[1:])
        tid = np.ascontiguousarray(tid, dtype=np.int32)
        anchor = np.ascontiguousarray(anchor, dtype=np.int32)
        range_max = np.ascontiguousarray(range_max, dtype=np.int32)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        b = ReadBatch(n, _ptr(bases), _ptr(off), _ptr(tid), _ptr(anchor), _ptr(range_max))
        rc = lib().im_realign_batch(self.h, C.byref(params), C.byref(b), _ptr(out))
        self._check(rc, allow=allow)
        return rc, out

    def depth_build(self, contig_len, seg_start, seg_len):
        seg_start = np.ascontiguousarray(seg_start, dtype=np.int32)
        seg_len = np.ascontiguousarray(seg_len, dtype=np.int32)
        self._check(lib().im_depth_build(self.h, contig_len, len(seg_start), _ptr(seg_start), _ptr(seg_len)))

    def depth_query(self, beg, end):
        beg = np.ascontiguousarray(beg, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        out = np.zeros(max(len(beg), 1), dtype=np.uint32)
        self._check(lib().im_depth_query(self.h, len(beg), _ptr(beg), _ptr(end), _ptr(out)))
        return out[:len(beg)]

    def support_batch(self, targets, queries):
        """targets / queries: lists of bytes.  Returns int32 [n,4]: subs, indels, aligned, status."""
        n = len(targets)
        t = np.frombuffer(b"".join(targets) + b"\0" * 8, dtype=np.uint8).copy()
        q = np.frombuffer(b"".join(queries) + b"\0" * 8, dtype=np.uint8).copy()
        to = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in targets], out=to[1:])
        qo = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in queries], out=qo[1:])
        out = np.zeros((max(n, 1), 4), dtype=np.int32)
        self._check(lib().im_support_batch(self.h, n, _ptr(t), _ptr(to), _ptr(q), _ptr(qo), _ptr(out)))
        return out[:n]

    def set_insert_ranges(self, names, range_max):
        """names in the order they entered the reference's insert-length table; range_max = range[1] of each."""
        n = len(names)
        arr = (C.c_char_p * max(n, 1))(*[x.encode() if isinstance(x, str) else x for x in names])
        rm = np.ascontiguousarray(range_max, dtype=np.int32)
        self._check(lib().im_set_insert_ranges(self.h, n, arr, _ptr(rm)))

    def depth_enable(self):
        self._check(lib().im_depth_enable(self.h))

    def depth_scan(self, tid, stream=None):
        self._check(lib().im_depth_scan(self.h, tid, self.stream if stream is None else stream))

    def depth_query_tid(self, tid, beg, end):
        beg = np.ascontiguousarray(beg, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        out = np.zeros(max(len(beg), 1), dtype=np.uint32)
        self._check(lib().im_depth_query_tid(self.h, tid, len(beg), _ptr(beg), _ptr(end), _ptr(out)))
        return out[:len(beg)]

    def cluster_sr(self, cls, b1, b2, marker=2**31 - 1, tie_desc=0):
        n = len(cls)
        cls = np.ascontiguousarray(cls, dtype=np.int32)
        b1 = np.ascontiguousarray(b1, dtype=np.int32)
        b2 = np.ascontiguousarray(b2, dtype=np.int32)
        order = np.zeros(max(n, 1), dtype=np.int32)
        first = np.zeros(max(n, 1), dtype=np.int32)
        count = np.zeros(max(n, 1), dtype=np.int32)
        used = np.zeros(max(n, 1), dtype=np.uint8)
        ncl = C.c_int32(0)
        self._check(lib().im_cluster_sr(self.h, n, _ptr(cls), _ptr(b1), _ptr(b2), marker, tie_desc,
                                        _ptr(order), _ptr(first), _ptr(count), _ptr(used), C.byref(ncl)))
        k = ncl.value
        return order[:n], first[:k], count[:k], used[:n], k


class Pipeline:
    """The device-resident hot path the product driver runs per batch of contigs (indelminer_amd/host):
    records -> triage (a1) -> realign (a2-a11) -> flush cuts -> group-by (a12).  Everything stays in HBM
    between the stages; this class only owns the buffers and issues the im_dev_* calls in order."""

    def __init__(self, ctx, n_records, raw_bytes, cap_cand, read_len_max=256, n_pe=0, n_flushes=64, want_depth=False,
                 qthreshold=10, ethreshold_vcfcheck=10, maxpedelsize=1000000, input_from=None):
        L = lib()
        self.ctx = ctx
        self.n_records = int(n_records)
        self.cap_cand = int(cap_cand)
        self.n_pe = int(n_pe)
        self.cap_bases = self.cap_cand * ((read_len_max + 3) // 4 * 4) + 64
        self.n_slots = self.cap_cand * MAX_EV + self.n_pe        # split-read slots, then the host's paired-read entries
        if input_from is not None:          # several output sets over one resident record buffer (bench.py's pipelined steps)
            self.d_raw, self.d_off = input_from.d_raw, input_from.d_off
        else:
            self.d_raw = DevBuf(ctx, raw_bytes + 64)
            self.d_off = DevBuf(ctx, 4 * (n_records + 1))
        self.d_bases = DevBuf(ctx, self.cap_bases)
        self.d_boff = DevBuf(ctx, 8 * self.cap_cand)
        self.d_len = DevBuf(ctx, 4 * self.cap_cand)
        self.d_tid = DevBuf(ctx, 4 * self.cap_cand)
        self.d_anchor = DevBuf(ctx, 4 * self.cap_cand)
        self.d_range = DevBuf(ctx, 4 * self.cap_cand)
        self.d_res = DevBuf(ctx, 512 * self.cap_cand)
        self.d_cls = DevBuf(ctx, 4 * self.n_slots)
        self.d_b1 = DevBuf(ctx, 4 * self.n_slots)
        self.d_b2 = DevBuf(ctx, 4 * self.n_slots)
        self.d_consumed = DevBuf(ctx, 4 * self.n_slots)
        self.d_cand_rec = DevBuf(ctx, 4 * self.cap_cand)
        self.d_counters = DevBuf(ctx, 64)
        self.d_class = DevBuf(ctx, max(n_records, 1))
        self.ts_bytes = L.im_dev_triage_scratch_bytes(n_records)
        self.d_ts = DevBuf(ctx, self.ts_bytes)
        ctx._check(L.im_dev_triage_scratch_init(ctx.h, n_records, self.d_ts.ptr, self.ts_bytes, ctx.stream))
        self.d_cut = DevBuf(ctx, 8 * max(n_flushes, 1))
        self.n_flushes = n_flushes
        self.d_order = DevBuf(ctx, 4 * self.n_slots)
        # {clusters, nodes, 0, 0} then the 16-byte cluster keys, contiguous: the unit a rank contributes to the all-gather
        self.d_clbuf = DevBuf(ctx, 16 + 16 * self.n_slots)
        self.d_counts = DevView(ctx, self.d_clbuf.ptr, 16)
        self.d_clkey = DevView(ctx, self.d_clbuf.ptr + 16, 16 * self.n_slots)
        self.d_clfirst = DevBuf(ctx, 4 * self.n_slots)
        self.d_clcount = DevBuf(ctx, 4 * self.n_slots)
        self.gs_bytes = L.im_dev_groupby_scratch_bytes(self.n_slots)
        self.d_gs = DevBuf(ctx, self.gs_bytes)
        ctx._check(L.im_dev_groupby_scratch_init(ctx.h, self.n_slots, self.d_gs.ptr, self.gs_bytes, ctx.stream))
        # the chip-wide form (im_dev_flush_groupby): its own table + the range-minimum tree over the flush list
        self.fg_bytes = L.im_dev_flushgroup_scratch_bytes(self.cap_cand * MAX_EV, max(n_flushes, 1))
        self.d_fg = DevBuf(ctx, self.fg_bytes)
        ctx._check(L.im_dev_flushgroup_scratch_init(ctx.h, self.cap_cand * MAX_EV, max(n_flushes, 1), self.d_fg.ptr, self.fg_bytes, ctx.stream))
        ctx._check(L.im_stream_sync(ctx.h, ctx.stream))
        self.batch = DevBatch(0, self.d_bases.ptr, self.d_boff.ptr, self.d_len.ptr, self.d_tid.ptr, self.d_anchor.ptr,
                              self.d_range.ptr, self.d_res.ptr, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr)
        self.cands = DevCands(self.batch, self.d_cand_rec.ptr, self.d_counters.ptr, self.d_class.ptr, self.cap_cand, self.cap_bases, None)
        # the same with the flush marks of new candidates cleared by the triage itself (bind_async: no fill per pass)
        self.cands_clear = DevCands(self.batch, self.d_cand_rec.ptr, self.d_counters.ptr, self.d_class.ptr, self.cap_cand, self.cap_bases,
                                    self.d_consumed.ptr)
        self.recs = DevRecords(self.n_records, self.d_raw.ptr, self.d_off.ptr, 0)
        self.tp = TriageParams(qthreshold, ethreshold_vcfcheck, maxpedelsize, 1 if want_depth else 0, 0, 0)
        self.P = params()
        self.n_cand = None

    def upload(self, raw, rec_off):
        self.d_raw.upload(raw)
        self.d_off.upload(rec_off)

    def set_pe(self, b1, b2):
        """the host's paired-read evidence: entries behind the split-read slots, class 2"""
        n = len(b1)
        assert n <= self.n_pe
        base = self.cap_cand * MAX_EV
        L = lib()
        for buf, arr in ((self.d_cls, np.full(n, 2, np.int32)), (self.d_b1, np.asarray(b1, np.int32)), (self.d_b2, np.asarray(b2, np.int32))):
            if n:
                self.ctx._check(L.im_dev_upload(self.ctx.h, buf.ptr + 4 * base, _ptr(np.ascontiguousarray(arr)), 4 * n))

    def triage(self, stream=None):
        """resets the running counters and the consumed marks, then classifies + compacts the records (asynchronous)"""
        L = lib()
        st = self.ctx.stream if stream is None else stream
        c = self.ctx._check
        c(L.im_dev_memset(self.ctx.h, self.d_counters.ptr, 0, 64, st))
        c(L.im_dev_memset(self.ctx.h, self.d_consumed.ptr, 0, 4 * self.n_slots, st))
        c(L.im_dev_memset(self.ctx.h, self.d_cut.ptr, 0xFF, 8 * self.n_flushes, st))
        c(L.im_dev_triage(self.ctx.h, C.byref(self.tp), C.byref(self.recs), C.byref(self.cands), self.d_ts.ptr, self.ts_bytes, st))

    def fetch_counts(self):
        """the one host synchronisation of a batch: candidates found (sizes the realign grid)"""
        self.sync()                         # the context's stream does not synchronise with the copy below by itself
        c = self.d_counters.download(np.int32, 8)
        self.n_cand = int(c[0])
        return c

    def realign(self, stream=None):
        st = self.ctx.stream if stream is None else stream
        self.batch.n = self.n_cand
        self.ctx._check(lib().im_dev_realign_keep(self.ctx.h, C.byref(self.P), C.byref(self.batch), st))

    def flush(self, k, cand_hi, pe_hi, marker, cand_lo=0, stream=None):
        """flush number k (0-based; its id is k + 1): pending split-read slots of candidates [cand_lo, cand_hi) and
        paired-read entries [0, pe_hi)"""
        st = self.ctx.stream if stream is None else stream
        base = self.cap_cand * MAX_EV
        self.ctx._check(lib().im_dev_flush_cut(self.ctx.h, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr,
                                               cand_lo * MAX_EV, cand_hi * MAX_EV, base, base + pe_hi, marker, k + 1,
                                               self.d_cut.ptr + 8 * k, st))

    def groupby(self, tie_desc=0, stream=None):
        st = self.ctx.stream if stream is None else stream
        n = self.n_cand * MAX_EV
        self.ctx._check(lib().im_dev_cluster_groupby(self.ctx.h, n, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr,
                                                     tie_desc, self.d_order.ptr, self.d_clkey.ptr, self.d_clfirst.ptr, self.d_clcount.ptr,
                                                     self.d_counts.ptr, self.d_gs.ptr, self.gs_bytes, st))

    # ---- the same three stages without a host round trip: the candidate count stays on the device ----
    @staticmethod
    def flush_descs(flushes):
        """im_flush_desc[] for [(rec0, rec1, pe_hi, marker)]: a change of rec0 opens a new contig; `last` = the contig's last flush"""
        n = len(flushes)
        desc = np.zeros((max(n, 1), 8), dtype=np.int32)
        for k, (rec0, rec1, pe_hi, marker) in enumerate(flushes):
            desc[k, :6] = (rec0, rec1, 0, pe_hi, marker, k + 1)
        k = 0
        while k < n:
            e = k
            while e + 1 < n and flushes[e + 1][0] == flushes[k][0]:
                e += 1
            desc[k:e + 1, 6] = e
            k = e + 1
        return desc[:n] if n else desc[:0]

    def flush_groupby(self, flushes, tie_desc=0, stream=None, cand_bound=None):
        """the flush list and the group-by chip-wide (im_dev_flush_groupby): three launches, no history"""
        st = self.ctx.stream if stream is None else stream
        desc = self.flush_descs(flushes)
        self.d_desc = DevBuf(self.ctx, max(desc.nbytes, 32)).upload(desc)
        self.ctx._check(lib().im_dev_flush_groupby(
            self.ctx.h, self.d_desc.ptr, len(flushes), self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr,
            self.d_cand_rec.ptr, self.d_counters.ptr, self.cap_cand if cand_bound is None else cand_bound, self.cap_cand * MAX_EV, self.n_pe, tie_desc,
            self.d_order.ptr, self.d_clkey.ptr, self.d_clfirst.ptr, self.d_clcount.ptr, self.d_counts.ptr, self.d_fg.ptr, self.fg_bytes, st))

    def bind_async(self, flushes, stream, tie_desc=0, grid_bound=None, one_launch_flushes=True, depth_tid=None, wide=False):
        """pre-binds one whole pass (triage -> realign -> flush cuts -> group-by) on `stream`; flushes =
        [(rec0, rec1, pe_hi, marker)] with record bounds.  Returns a list of (fn, args) to call in order.
        wide: the flush list + group-by as im_dev_flush_groupby (three chip-wide launches) instead of the
        one-workgroup flush list + the four group-by launches."""
        L = lib()
        h = self.ctx.h
        base = self.cap_cand * MAX_EV
        self.batch_bound = DevBatch(min(self.cap_cand, grid_bound or self.cap_cand), self.d_bases.ptr, self.d_boff.ptr, self.d_len.ptr,
                                    self.d_tid.ptr, self.d_anchor.ptr, self.d_range.ptr, self.d_res.ptr, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr)
        # a pass over one resident chunk opens a new batch with its one triage call: tp.restart instead of a counters memset
        self.tp_restart = TriageParams(self.tp.qthreshold, self.tp.ethreshold_vcfcheck, self.tp.maxpedelsize, self.tp.want_depth, 0, 1)
        calls = []
        if depth_tid is not None:           # the contig goes through triage again: its run of the depth array starts from zeros
            calls.append((L.im_depth_reset, (h, depth_tid, stream)))
        if not one_launch_flushes and not wide:
            calls += [(L.im_dev_memset, (h, self.d_consumed.ptr, 0, 4 * self.n_slots, stream)),
                      (L.im_dev_memset, (h, self.d_cut.ptr, 0xFF, 8 * self.n_flushes, stream))]
        calls += [(L.im_dev_triage, (h, C.byref(self.tp_restart), C.byref(self.recs), C.byref(self.cands_clear if (one_launch_flushes and not wide) else self.cands),
                                     self.d_ts.ptr, self.ts_bytes, stream)),
                  (L.im_dev_realign_n, (h, C.byref(self.P), C.byref(self.batch_bound), self.d_counters.ptr, 1, stream))]
        self.realign_call_index = len(calls) - 1
        self.triage_call_index = len(calls) - 2
        if depth_tid is not None:           # the contig's records are all in: difference array -> depths (DP= queries of the replay)
            calls.append((L.im_depth_scan, (h, depth_tid, stream)))
        if wide:
            desc = self.flush_descs(flushes)
            self.d_desc = DevBuf(self.ctx, max(desc.nbytes, 32)).upload(desc)
            calls.append((L.im_dev_flush_groupby, (h, self.d_desc.ptr, len(flushes), self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr,
                                                   self.d_cand_rec.ptr, self.d_counters.ptr, self.batch_bound.n, base, self.n_pe, tie_desc,
                                                   self.d_order.ptr, self.d_clkey.ptr, self.d_clfirst.ptr, self.d_clcount.ptr, self.d_counts.ptr,
                                                   self.d_fg.ptr, self.fg_bytes, stream)))
            return calls
        if one_launch_flushes:
            desc = self.flush_descs(flushes)
            self.d_desc = DevBuf(self.ctx, max(desc.nbytes, 32)).upload(desc)
            calls.append((L.im_dev_flush_cuts, (h, self.d_desc.ptr, len(flushes), self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr,
                                                self.d_cand_rec.ptr, self.d_counters.ptr, self.cap_cand, base, self.n_pe, stream)))
        else:
            for k, (rec0, rec1, pe_hi, marker) in enumerate(flushes):
                calls.append((L.im_dev_flush_cut_rec, (h, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr, self.d_consumed.ptr, rec0, rec1,
                                                       self.d_cand_rec.ptr, self.d_counters.ptr, self.cap_cand, base, base + pe_hi, marker, k + 1,
                                                       self.d_cut.ptr + 8 * k, stream)))
        calls.append((L.im_dev_cluster_groupby_n, (h, self.batch_bound.n * MAX_EV, self.d_counters.ptr, self.d_cls.ptr, self.d_b1.ptr, self.d_b2.ptr,
                                                   self.d_consumed.ptr, tie_desc, self.d_order.ptr, self.d_clkey.ptr, self.d_clfirst.ptr,
                                                   self.d_clcount.ptr, self.d_counts.ptr, self.d_gs.ptr, self.gs_bytes, stream)))
        return calls

    def sync(self, stream=None):
        self.ctx._check(lib().im_stream_sync(self.ctx.h, self.ctx.stream if stream is None else stream))

    def clusters(self):
        """(key[n,4] = flush id, class, b1, b2; first; count; order) after groupby + sync"""
        k, nodes = (int(x) for x in self.d_counts.download(np.int32, 2))
        key = self.d_clkey.download(np.int32, 4 * max(k, 1)).reshape(-1, 4)[:k]
        return key, self.d_clfirst.download(np.int32, max(k, 1))[:k], self.d_clcount.download(np.int32, max(k, 1))[:k], \
            self.d_order.download(np.int32, max(nodes, 1))[:nodes]


COMM_ID_BYTES = 128


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().im_comm_unique_id(buf)
    if rc != IM_OK:
        raise IMError(rc, lib().im_comm_last_error().decode())
    return buf.raw


class Comm:
    """RCCL communicator over the C ABI (one all-gather of cluster lists per step)."""

    def __init__(self, ctx, id_bytes, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        rc = lib().im_comm_init(ctx.h, id_bytes, rank, world, C.byref(h))
        if rc != IM_OK:
            raise IMError(rc, lib().im_comm_last_error().decode())
        self.h = h.value

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank, stream):
        rc = lib().im_comm_allgather(self.h, send_ptr, recv_ptr, bytes_per_rank, stream)
        if rc != IM_OK:
            raise IMError(rc, lib().im_comm_last_error().decode())

    def close(self):
        if self.h:
            lib().im_comm_destroy(self.h)
            self.h = None


def params(klength=6, numgaps=0, maxdelsize=1000, ethreshold=10):
    return Params(klength, numgaps, maxdelsize, ethreshold)
