"""The round-1 realign-only shard (host-prepared candidate batches, im_dev_realign + the histogram cluster path) that the
probes in this directory were written against.  Not part of bench.py or the product."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indelminer_amd import capi  # noqa: E402

PIPELINE_DEPTH = int(os.environ.get("IM_BENCH_DEPTH", "4"))


class Shard:
    """Device-resident state of one rank's shard."""

    def __init__(self, ctx, ref, cand, read_len):
        self.ctx = ctx
        n = len(cand["index"])
        self.n = n
        L = read_len
        stride = (L + 3) // 4 * 4
        bases = np.zeros((n, stride), dtype=np.uint8)
        bases[:, :L] = cand["bases"]
        flat = np.concatenate([bases.reshape(-1), np.zeros(16, np.uint8)])
        self.d_bases = capi.DevBuf(ctx, flat.nbytes).upload(flat)
        self.d_off = capi.DevBuf(ctx, 8 * n).upload(np.arange(n, dtype=np.int64) * stride)
        self.d_len = capi.DevBuf(ctx, 4 * n).upload(np.full(n, L, np.int32))
        self.d_tid = capi.DevBuf(ctx, 4 * n).upload(np.zeros(n, np.int32))
        self.d_anchor = capi.DevBuf(ctx, 4 * n).upload(cand["anchor"].astype(np.int32))
        self.d_range = capi.DevBuf(ctx, 4 * n).upload(cand["range_max"].astype(np.int32))
        # PIPELINE_DEPTH sets of realign outputs: while the cluster kernels of step i read set i % depth on the
        # cluster stream, the realign kernels of the following steps write the other sets on the context's stream.
        self.cap = n * capi.MAX_EV               # evidence slots: IM_MAX_EV per read
        cap = self.cap
        self.sets = []
        for _ in range(PIPELINE_DEPTH):
            self.sets.append({"res": capi.DevBuf(ctx, 512 * n), "cls": capi.DevBuf(ctx, 4 * cap),
                              "b1": capi.DevBuf(ctx, 4 * cap), "b2": capi.DevBuf(ctx, 4 * cap),
                              "realigned": capi.Event(ctx), "clustered": capi.Event(ctx)})
        self.d_res, self.d_cls, self.d_b1, self.d_b2 = (self.sets[0][k] for k in ("res", "cls", "b1", "b2"))
        self.cluster_stream = capi.new_stream(ctx)
        self.k = 0                               # steps issued
        self.d_order = capi.DevBuf(ctx, 4 * cap)
        self.d_first = capi.DevBuf(ctx, 4 * cap)
        self.d_count = capi.DevBuf(ctx, 4 * cap)
        self.d_used = capi.DevBuf(ctx, cap)
        self.d_counts = capi.DevBuf(ctx, 64)
        # multi-kernel path (more live evidence than one workgroup sorts in LDS)
        L_ = capi.lib()
        self.d_src = capi.DevBuf(ctx, 4 * cap)
        self.d_nout = capi.DevBuf(ctx, 64)
        self.d_dcls = capi.DevBuf(ctx, 4 * cap)
        self.d_db1 = capi.DevBuf(ctx, 4 * cap)
        self.d_db2 = capi.DevBuf(ctx, 4 * cap)
        self.gs_bytes = L_.im_dev_gather_scratch_bytes(n)
        self.d_gs = capi.DevBuf(ctx, self.gs_bytes)
        self.cs_bytes = L_.im_dev_cluster_scratch_bytes(cap)
        self.d_cs = capi.DevBuf(ctx, self.cs_bytes)
        for st_ in self.sets:
            st_["batch"] = capi.DevBatch(n, self.d_bases.ptr, self.d_off.ptr, self.d_len.ptr, self.d_tid.ptr,
                                         self.d_anchor.ptr, self.d_range.ptr, st_["res"].ptr,
                                         st_["cls"].ptr, st_["b1"].ptr, st_["b2"].ptr)
        self.batch = self.sets[0]["batch"]
        self.hs_bytes = L_.im_dev_cluster_hist_scratch_bytes(cap)
        self.d_hs = capi.DevBuf(ctx, self.hs_bytes)
        ctx._check(L_.im_dev_cluster_hist_init(ctx.h, cap, self.d_hs.ptr, self.hs_bytes, ctx.stream))
        ctx._check(L_.im_stream_sync(ctx.h, ctx.stream))           # the cluster stream starts after the table is ready
        self.P = capi.params()
        self.small = True                        # breakpoint-histogram cluster path; cleared if it overflows
        # multi-GPU: per-shard cluster list (16 B records) and the gathered lists of all ranks
        self.tid = 0
        self.rec_cap = 16384
        self.comm = None
        self.d_gather = None

    def set_rec_cap(self, cap):
        self.rec_cap = int(cap)         # before attach_comm, which allocates the record / gather buffers

    def attach_comm(self, comm):
        """The all-gather gets a stream of its own and every buffer set its own record / gather buffers, so that
        the collective of step i overlaps the cluster kernels of step i + 1 as well as the realign kernels."""
        self.comm = comm
        self.comm_stream = capi.new_stream(self.ctx)
        for cur in self.sets:
            cur["recs"] = capi.DevBuf(self.ctx, 16 * self.rec_cap)
            cur["gather"] = capi.DevBuf(self.ctx, 16 * self.rec_cap * comm.world)
            cur["recorded"] = capi.Event(self.ctx)
        self.d_gather = self.sets[0]["gather"]

    def _bind(self):
        """Pre-bound foreign calls of one step per buffer set: the step loop is host-issue bound otherwise
        (a ctypes call with a dozen arguments costs microseconds; profiles/overlap_probe.py)."""
        L_ = capi.lib()
        ctx = self.ctx
        st, sc = ctx.stream, self.cluster_stream
        for cur in self.sets:
            calls = []
            if self.small:
                calls.append((L_.im_dev_cluster_hist, (ctx.h, self.cap, cur["cls"].ptr, cur["b1"].ptr, cur["b2"].ptr,
                                                       2**31 - 1, 0, self.d_order.ptr, self.d_first.ptr, self.d_count.ptr,
                                                       self.d_used.ptr, self.d_counts.ptr, self.d_hs.ptr, self.hs_bytes, sc)))
            else:
                calls.append((L_.im_dev_gather_evidence, (ctx.h, cur["res"].ptr, self.n, self.d_dcls.ptr, self.d_db1.ptr,
                                                          self.d_db2.ptr, self.d_src.ptr, self.cap, self.d_nout.ptr,
                                                          self.d_gs.ptr, self.gs_bytes, sc)))
                calls.append((L_.im_dev_cluster_sr, (ctx.h, self.cap, self.d_nout.ptr, self.d_dcls.ptr, self.d_db1.ptr, self.d_db2.ptr,
                                                     2**31 - 1, 0, self.d_order.ptr, self.d_first.ptr, self.d_count.ptr,
                                                     self.d_used.ptr, self.d_counts.ptr, self.d_cs.ptr, self.cs_bytes, sc)))
            if self.comm is not None:
                # the one collective of the path: all-gather of the per-shard cluster lists (RCCL over xGMI)
                src = (cur["cls"], cur["b1"], cur["b2"]) if self.small else (self.d_dcls, self.d_db1, self.d_db2)
                calls.append((L_.im_dev_cluster_records, (ctx.h, self.tid, self.d_counts.ptr, self.d_order.ptr, self.d_first.ptr,
                                                          self.d_count.ptr, src[0].ptr, src[1].ptr, src[2].ptr,
                                                          cur["recs"].ptr, self.rec_cap, sc)))
                cur["to_comm_args"] = (cur["recorded"].h, sc, self.comm_stream)
                cur["done_args_comm"] = (cur["clustered"].h, self.comm_stream)
            cur["realign_args"] = (ctx.h, C.byref(self.P), C.byref(cur["batch"]), st)
            cur["cluster_calls"] = calls
            cur["follow_args"] = (cur["realigned"].h, st, sc)
            cur["done_args"] = (cur["clustered"].h, sc)
        self._bound = (self.small, self.comm)

    def step(self, timer=None):
        """One pass of the hot path over the resident batch: realign on the context's stream, the cluster
        kernels (and the all-gather) behind it on the cluster stream, so that they overlap the NEXT step's
        realign kernel.  The host throttles on the set's previous use; the realign stream carries no wait."""
        if getattr(self, "_bound", None) != (self.small, self.comm):
            self._bind()
        L_ = capi.lib()
        cur = self.sets[self.k % PIPELINE_DEPTH]
        if self.k >= PIPELINE_DEPTH:
            cur["clustered"].sync()     # step k - PIPELINE_DEPTH has finished reading this set (host waits, not the GPU)
        if timer is not None:
            timer.start(self.ctx.stream)
        rc = L_.im_dev_realign(*cur["realign_args"])
        if timer is not None:
            timer.stop(self.ctx.stream)
        rc = rc or L_.im_stream_follow(*cur["follow_args"])
        for fn, args in cur["cluster_calls"]:
            rc = rc or fn(*args)
        if self.comm is not None:
            rc = rc or L_.im_stream_follow(*cur["to_comm_args"])
            self.comm.allgather(cur["recs"].ptr, cur["gather"].ptr, 16 * self.rec_cap, self.comm_stream)
            rc = rc or L_.im_event_record(*cur["done_args_comm"])      # the set is free again once its gather is done
            self.d_gather = cur["gather"]
        else:
            rc = rc or L_.im_event_record(*cur["done_args"])
        if rc:
            self.ctx._check(rc)
        self.d_res = cur["res"]
        self.k += 1

    def sync(self):
        self.ctx._check(capi.lib().im_stream_sync(self.ctx.h, self.ctx.stream))
        self.ctx._check(capi.lib().im_stream_sync(self.ctx.h, self.cluster_stream))
        if self.comm is not None:
            self.ctx._check(capi.lib().im_stream_sync(self.ctx.h, self.comm_stream))

    def results(self):
        return self.d_res.download(capi.RESULT_DTYPE, self.n)

    def clusters(self):
        if self.small:
            c = self.d_counts.download(np.int32, 2)
            return int(c[0]), int(c[1])
        return int(self.d_counts.download(np.int32, 1)[0]), int(self.d_nout.download(np.int32, 1)[0])
