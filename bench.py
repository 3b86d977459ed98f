#!/usr/bin/env python3
"""bench.py -- reads/sec through split-read realign + cluster on MI355X.

One "step" = one pass of the hot path over one resident batch of DELIVERED BAM RECORDS, exactly the
sequence of C-ABI calls the product driver (indelminer_amd/host, run_pipeline) issues per batch:
  im_depth_reset       the contig's run of the depth array (the pass is repeated on the same contig; the product zeroes once)
  im_dev_triage        every record: fetch_func's candidate rules, base decode + revcomp, CIGAR evidence, pileup depth scatter (a1, a16)
  im_depth_scan        the contig's depths for the DP= queries (a16)
  im_dev_realign_n     every candidate: K1-K4, evidence slots (a2-a11)
  im_dev_flush_groupby every READCHUNK flush point + the end-of-contig flush (a12, node selection) and the split-read clusters of
                       all flushes (a12) in three chip-wide launches (IM_BENCH_FLUSH=seq: the one-workgroup flush list + four group-by launches)
After the timed region every buffer set is checked against the pass run alone and the depths against the pileup rule;
at N > 1 (or IM_BENCH_FORCE_COMM=1) the product CLI is also run on all ranks (end_to_end_multi_gpu).
Inputs (reference, the records as the BAM file holds them) are resident in HBM before the timed region
starts; `value` counts the records the timed kernels read -- all of them.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): synthetic 1 Mb contig, 100 bp PE reads at
30x with seeded 1-50 bp indels, per GPU (weak scaling: every rank owns one such
contig, seeded by its rank; the path shards by contig with no data-path
collective in the timed step).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from indelminer_amd import capi, rawrec, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PIPELINE_DEPTH = int(os.environ.get("IM_BENCH_DEPTH", "4"))   # sets of realign output buffers in flight
MIN_TIMED_S = 0.12              # the --steps long region is repeated until at least this much time has been measured (and at least
MIN_REGIONS, MAX_REGIONS = 5, 400   # MIN_REGIONS times): ms_per_step is the MEDIAN region's, the spread is reported beside it
BRACKET_STEPS = 96              # after the timed regions: this many overlapped steps of which every fourth has ONE launch bracketed by HIP
                                # events on its stream (realign / triage alternating); those steps are issued call by call, not as a graph
FLUSH_WIDE = os.environ.get("IM_BENCH_FLUSH", "wide") == "wide"   # im_dev_flush_groupby (3 chip-wide launches) or the sequential flush list + 4 group-by launches


def measured_traffic():
    """(HBM bytes per realign launch, source file) from the newest PMC passes committed under profiles/ (FETCH_SIZE x 2
    per the gfx950 correction + WRITE_SIZE, collected with separate --pmc runs of this same command by
    profiles/collect.sh); counters cannot be read from inside this process."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_traffic.json")) if "shard3" not in f)
    try:
        with open(files[-1]) as fh:
            return int(json.load(fh)["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def measured_issue():
    """What binds the realign kernel is instruction issue, not HBM: per-read instruction counts, the share of the SIMDs' issue
    cycles its VALU instructions take and the LDS bank-conflict share, from the SQ counter passes of the same committed set
    (profiles/traffic_json.py writes them into the traffic file next to the HBM bytes)."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_traffic.json")) if "shard3" not in f)
    try:
        with open(files[-1]) as fh:
            return json.load(fh).get("issue")
    except Exception:
        return None


def algorithmic_bytes(res):
    """per candidate, sum over its band searches of (window bytes +
    read-piece bytes) at 1 B/base, + 32 B per band alignment + 64 B per evidence record."""
    nb = res["n_band"].astype(np.int64)
    band = res["band"]
    w = band["win_bytes"].astype(np.int64) + band["piece_bytes"].astype(np.int64)
    mask = np.arange(2)[None, :] < nb[:, None]
    total = int((w * mask).sum()) + 32 * int(nb.sum())
    total += 64 * int(np.where(res["status"] == 1, res["n_ev"], 0).sum())
    return total


READCHUNK = 100000          # src/indelminer.c:28


def flush_schedule(rd):
    """The host driver's part of the walk for a one-contig shard, restated on the simulator's columns: which records
    are counted (src/indelminer.c:348-366), where the READCHUNK flush points fall (617) and their markers
    (find_marker over the waiting first mates, 211-233, then min with the record's position, 623), and the
    paired-read evidence (516-615) as (b1, b2, completing record).  Returns (flushes [(rec0, rec1, pe_hi, marker)], pe_b1, pe_b2)."""
    f = rd.flag
    counted = ((f & 0xF00) == 0) & ((f & 0x1) != 0)
    aligned = (f & 0x4) == 0
    mate_al = (f & 0x8) == 0
    disc = counted & aligned & mate_al & ((f & 0x2) == 0) & (np.abs(rd.isize) > rd.range_max) & (np.abs(rd.isize) < 1000000) & \
        (((f & 0x10) != 0) != ((f & 0x20) != 0))
    span = (rd.cig_len * np.isin(rd.cig_op, (0, 2, 7, 8)) * (np.arange(rd.cig_op.shape[1])[None, :] < rd.ncig[:, None])).sum(1)
    waiting = {}
    pe = []
    events = {}                                   # record index -> waiting-set change, only where discordant pairs exist
    for i in np.nonzero(disc)[0]:
        pid = int(rd.pair_id[i])
        if rd.pos[i] < rd.mpos[i]:
            waiting[pid] = (int(rd.pos[i]), int(rd.pos[i] + span[i]))
            events[int(i)] = ("add", pid, int(rd.pos[i]))
        elif pid in waiting:
            st, en = waiting.pop(pid)
            pe.append((en, int(rd.pos[i]), int(i)))
            events[int(i)] = ("del", pid, st)
    cum = np.cumsum(counted)
    marks = np.nonzero(counted & (cum % READCHUNK == 0))[0]
    flushes = []
    live = {}
    ev_idx = sorted(events)
    p = 0
    pe_rec = np.array([x[2] for x in pe], dtype=np.int64)
    for m in marks:
        while p < len(ev_idx) and ev_idx[p] <= m:
            kind, pid, st = events[ev_idx[p]]
            if kind == "add":
                live[pid] = st
            else:
                live.pop(pid, None)
            p += 1
        marker = min([int(rd.pos[m])] + list(live.values()))
        flushes.append((0, int(m) + 1, int((pe_rec <= m).sum()), marker))
    flushes.append((0, rd.n, len(pe), 2**31 - 1))
    return flushes, np.array([x[0] for x in pe], np.int32), np.array([x[1] for x in pe], np.int32)


class PipeStep:
    """Device-resident state of one rank's shard for the product pipeline: the delivered records of the shard in
    HBM, and DEPTH sets of output buffers so that consecutive steps overlap on DEPTH streams (the host waits for a
    set's previous use; nothing else synchronises inside the timed region)."""

    def __init__(self, ctx, rd, depth):
        self.ctx = ctx
        raw, off = rawrec.records(rd, qual=os.environ.get("IM_BENCH_KEEP_QUAL") == "1")     # as the product's walkers deliver them: without base qualities
        self.n_records = rd.n
        self.record_bytes = int(len(raw))
        flushes, pe_b1, pe_b2 = flush_schedule(rd)
        self.flushes = flushes
        cap = max(4096, rd.n // 8)
        ctx.depth_enable()
        self.sets = []
        self.raw0, self.off = raw, off
        at = (off[:-1] // 4).astype(np.int64)
        for k in range(depth):
            # buffer set k works on contig k of the context (identical copies of the shard's contig): its own depth run,
            # as consecutive groups of the product work on different contigs
            rk = raw.copy()
            r32 = rk[:len(rk) // 4 * 4].view(np.int32)
            for word in (0, 5):                 # refID, next_refID
                v = r32[at + word]
                r32[at + word] = np.where(v >= 0, k, v)
            pipe = capi.Pipeline(ctx, rd.n, len(raw), cap_cand=cap, read_len_max=rd.read_len, n_pe=max(len(pe_b1), 1),
                                 n_flushes=len(flushes), want_depth=True)
            pipe.upload(rk, off)
            pipe.set_pe(pe_b1, pe_b2)
            stream = capi.new_stream(ctx)
            self.sets.append({"pipe": pipe, "stream": stream, "done": capi.Event(ctx), "raw_host": rk,
                              "calls": pipe.bind_async(flushes, stream, grid_bound=cap, depth_tid=k, wide=FLUSH_WIDE), "graph": None, "tail": []})
        self.k = 0

    def depth_check(self, rd, clen, n_query=2000, seed=5):
        """DP= range sums of every buffer set's contig against the pileup rule applied to the records on the host"""
        from tests.support import oraclebind as ob
        want = ob.depth_of(self.raw0, self.off, 0, clen).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(want)])
        rng = np.random.default_rng(seed)
        beg = rng.integers(0, clen - 1, n_query).astype(np.int32)
        end = np.minimum(beg + rng.integers(1, 2000, n_query), clen).astype(np.int32)
        exp = (cs[end] - cs[beg]).astype(np.uint32)
        return all(np.array_equal(self.ctx.depth_query_tid(k, beg, end), exp) for k in range(len(self.sets)))

    def capture(self):
        """One HIP graph per buffer set: the step's launches (3 fills, 4 triage kernels, realign, the flush list, 4 group-by
        kernels + its table fill) are recorded once and replayed with one host call, like the product's per-batch sequence
        would be if the host, not the GPU, were the limit.  Anything appended to a set's `tail` (the all-gather) follows the
        graph on the same stream."""
        L_ = capi.lib()
        for cur in self.sets:
            self.ctx._check(L_.im_capture_begin(self.ctx.h, cur["stream"]))
            rc = 0
            for fn, args in cur["calls"]:
                rc = rc or fn(*args)
            g = C.c_void_p()
            rc2 = L_.im_capture_end(self.ctx.h, cur["stream"], C.byref(g))
            self.ctx._check(rc)
            self.ctx._check(rc2)
            cur["graph"] = g.value

    def step(self, timer=None, timer_at=None):
        cur = self.sets[self.k % len(self.sets)]
        if self.k >= len(self.sets):
            cur["done"].sync()
        rc = 0
        if timer is None and cur["graph"] is not None:
            rc = capi.lib().im_graph_launch(cur["graph"], cur["stream"])
        else:
            for j, (fn, args) in enumerate(cur["calls"]):
                if timer is not None and j == timer_at:
                    timer.start(cur["stream"])
                    rc = rc or fn(*args)
                    timer.stop(cur["stream"])
                else:
                    rc = rc or fn(*args)
        for fn, args in cur["tail"]:
            rc = rc or fn(*args)
        if rc:
            self.ctx._check(rc)
        cur["done"].record(cur["stream"])
        self.last = cur
        self.k += 1

    def sync(self):
        for st_ in self.sets:
            self.ctx._check(capi.lib().im_stream_sync(self.ctx.h, st_["stream"]))

    def measure_with_uploads(self, steps):
        """The same overlapped passes with the step's records CROSSING PCIe first: every step uploads its chunk (records + offsets)
        from pinned host memory on its stream, then runs the pass -- what the product's walkers do.  Returns seconds per step."""
        L_ = capi.lib()
        import ctypes
        pinned = []
        for cur in self.sets:
            p = cur["pipe"]
            hr, ho = ctypes.c_void_p(), ctypes.c_void_p()
            self.ctx._check(L_.im_host_alloc(self.ctx.h, len(self.raw0) + 64, ctypes.byref(hr)))
            self.ctx._check(L_.im_host_alloc(self.ctx.h, 4 * len(self.off), ctypes.byref(ho)))
            ctypes.memmove(hr.value, np.ascontiguousarray(cur["raw_host"]).ctypes.data, len(self.raw0))
            ctypes.memmove(ho.value, np.ascontiguousarray(self.off).ctypes.data, 4 * len(self.off))
            pinned.append((hr, ho))
            cur["upload"] = [(L_.im_dev_upload_async, (self.ctx.h, p.d_raw.ptr, hr.value, len(self.raw0), cur["stream"])),
                             (L_.im_dev_upload_async, (self.ctx.h, p.d_off.ptr, ho.value, 4 * len(self.off), cur["stream"]))]
        self.sync()
        t = time.perf_counter()
        for _ in range(steps):
            cur = self.sets[self.k % len(self.sets)]
            if self.k >= len(self.sets):
                cur["done"].sync()
            rc = 0
            for fn, args in cur["upload"]:
                rc = rc or fn(*args)
            rc = rc or (L_.im_graph_launch(cur["graph"], cur["stream"]) if cur["graph"] is not None else 0)
            if cur["graph"] is None:
                for fn, args in cur["calls"]:
                    rc = rc or fn(*args)
            if rc:
                self.ctx._check(rc)
            cur["done"].record(cur["stream"])
            self.last = cur
            self.k += 1
        self.sync()
        dt = (time.perf_counter() - t) / steps
        for hr, ho in pinned:
            L_.im_host_free(self.ctx.h, hr); L_.im_host_free(self.ctx.h, ho)
        return dt

    def quiet_launch_times(self, n=12):
        """HIP events around the realign launch and around the triage launches with NOTHING else on the device: the pass
        issued call by call on one buffer set, the stream drained before every bracket.  This is the kernel's own duration
        (what rocprofv3's kernel trace reports); the brackets taken inside the overlapped timed region also contain the wait
        for a dispatch slot behind the other streams' launches."""
        cur = self.sets[0]
        p = cur["pipe"]
        L_ = capi.lib()
        out = {p.realign_call_index: [], p.triage_call_index: []}
        tm = capi.Timer(self.ctx)
        for _ in range(n):
            for j, (fn, args) in enumerate(cur["calls"]):
                if j in out:
                    self.ctx._check(L_.im_stream_sync(self.ctx.h, cur["stream"]))
                    tm.start(cur["stream"])
                    self.ctx._check(fn(*args))
                    tm.stop(cur["stream"])
                    out[j].append(tm.elapsed_ms())
                else:
                    self.ctx._check(fn(*args))
            self.ctx._check(L_.im_stream_sync(self.ctx.h, cur["stream"]))
        return float(np.mean(out[p.realign_call_index][1:])), float(np.mean(out[p.triage_call_index][1:]))

    def digest(self, st_):
        """what a pass left in one buffer set, in a form that does not depend on hash-table order: the candidate count, every
        read's status / evidence, the consumed marks and the clusters as a sorted list of (key, members)"""
        p = st_["pipe"]
        c = p.fetch_counts()
        nc = int(c[0])
        res = p.d_res.download(capi.RESULT_DTYPE, max(nc, 1))[:nc]
        key, first, count, order = p.clusters()
        cl = sorted((tuple(int(x) for x in key[i]), tuple(int(x) for x in order[first[i]:first[i] + count[i]])) for i in range(len(key)))
        cons = p.d_consumed.download(np.int32, p.n_slots)
        import hashlib
        h = hashlib.md5()
        h.update(np.ascontiguousarray(res["status"]).tobytes()); h.update(np.ascontiguousarray(res["n_ev"]).tobytes())
        h.update(np.ascontiguousarray(res["ev"]["b1"]).tobytes()); h.update(np.ascontiguousarray(res["ev"]["b2"]).tobytes())
        h.update(cons[:nc * capi.MAX_EV].tobytes()); h.update(cons[p.cap_cand * capi.MAX_EV:p.cap_cand * capi.MAX_EV + p.n_pe].tobytes()); h.update(repr(cl).encode())
        return nc, len(cl), h.hexdigest()

    def results(self):
        p = self.last["pipe"]
        c = p.fetch_counts()
        return c, p.d_res.download(capi.RESULT_DTYPE, max(int(c[0]), 1))[:int(c[0])], p.d_counts.download(np.int32, 2)


def copy_peak_gbs(ctx, nbytes=1 << 30, reps=5):
    """device-to-device copy bandwidth of this GPU (read + write bytes per second), the measured figure beside the 8 TB/s spec"""
    a = capi.DevBuf(ctx, nbytes); b = capi.DevBuf(ctx, nbytes)
    L_ = capi.lib()
    t = capi.Timer(ctx)
    ctx._check(L_.im_dev_copy_async(ctx.h, b.ptr, a.ptr, nbytes, ctx.stream))
    best = 1e9
    for _ in range(reps):
        t.start(ctx.stream)
        ctx._check(L_.im_dev_copy_async(ctx.h, b.ptr, a.ptr, nbytes, ctx.stream))
        t.stop(ctx.stream)
        best = min(best, t.elapsed_ms())
    a.free(); b.free()
    return 2.0 * nbytes / (best * 1e-3) / 1e9


def h2d_pinned_gbs(ctx, nbytes=256 << 20, reps=4):
    """host -> device copy rate from pinned memory (what the product's chunk uploads get): the PCIe share of a pass whose
    records start on the host.  Reported beside `value`, which is measured with the records already resident."""
    L_ = capi.lib()
    hp = C.c_void_p()
    ctx._check(L_.im_host_alloc(ctx.h, nbytes, C.byref(hp)))
    d = capi.DevBuf(ctx, nbytes)
    t = capi.Timer(ctx)
    best = 1e9
    for _ in range(reps + 1):
        t.start(ctx.stream)
        ctx._check(L_.im_dev_upload_async(ctx.h, d.ptr, hp, nbytes, ctx.stream))
        t.stop(ctx.stream)
        best = min(best, t.elapsed_ms())
    d.free()
    L_.im_host_free(ctx.h, hp)
    return nbytes / (best * 1e-3) / 1e9


def cpu_reference_baseline(ref, cand, read_len, n_reads_total, budget_s=12.0):
    """Times the REAL reference's attempt_pe_alignment (oracle/_ref, compiled in place from the
    reference sources) on a bounded sample of the same candidate batch, single thread."""
    lib_path = os.path.join(ROOT, "oracle", "_ref", "librefbatch.so")
    n = len(cand["index"])
    if not os.path.exists(lib_path):
        return None, None
    L = C.CDLL(lib_path)
    L.rb_run.restype = C.c_double
    L.rb_set_params(6, 0, 1000, 10)
    contig = C.create_string_buffer(ref.tobytes())
    seqs = (C.c_char_p * 1)(C.cast(contig, C.c_char_p))
    # probe 200 reads to size the sample for ~budget seconds
    def run(m):
        bases = np.ascontiguousarray(cand["bases"][:m]).reshape(-1)
        off = (np.arange(m + 1, dtype=np.int64) * read_len)
        tid = np.zeros(m, np.int32)
        anchor = np.ascontiguousarray(cand["anchor"][:m], dtype=np.int32)
        rng = np.ascontiguousarray(cand["range_max"][:m], dtype=np.int32)
        out = np.zeros(4 * m, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        t = L.rb_run(seqs, C.c_int32(m), p(bases), p(off), p(tid), p(anchor), p(rng), p(out))
        return t, out.reshape(m, 4)
    t_probe, _ = run(min(200, n))
    per = t_probe / min(200, n)
    m = int(min(n, max(200, budget_s / max(per, 1e-9))))
    t, out = run(m)
    frac = n / float(n_reads_total)             # candidates per delivered read
    reads_equiv = m / frac
    info = {"value": reads_equiv / t, "unit": "reads/s", "cores": 1, "kind": "reference",
            "sample": "%d of %d candidate reads (= %.0f delivered reads at %.2f%% candidates) through the reference's "
                      "attempt_pe_alignment compiled from its own sources, %.2f s, 1 thread; realign only, BAM decode "
                      "and clustering not included" % (m, n, reads_equiv, 100 * frac, t),
            "candidates_per_s": m / t}
    return info, out


def end_to_end(refs, rd):
    """Whole programs on the bench workload written out as BAM: the reference binary (compiled in
    place from its sources, oracle/_ref) and the product driver (indelminer_amd/indelminer, GPU),
    same flags, VCF compared byte for byte.  Reported beside the kernel-level numbers; never `value`."""
    import subprocess
    import tempfile
    from indelminer_amd import bamwrite, build
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    prod = build.HOST_BIN
    if not os.path.exists(prod):
        return None
    with tempfile.TemporaryDirectory() as td:
        contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
        bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
        rawrec.write_bam_fast(td + "/aln.bam", contigs, rd, level=6)
        open(td + "/cfg.txt", "w").write("IL generic 300 %d\n" % rd.range_max)
        cmd = ["-i", "cfg.txt", "ref.fa", "s=aln.bam"]
        out = {"reads": int(rd.n)}
        walls = []
        for _ in range(2):                      # the first run also pages the binary, the libraries and the inputs in
            t = time.perf_counter()
            p = subprocess.run([prod] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            walls.append(time.perf_counter() - t)
        out["product_wall_s"] = min(walls)
        out["product_wall_s_runs"] = walls
        out["product_rc"] = p.returncode
        out["vcf_records"] = sum(1 for l in p.stdout.splitlines() if not l.startswith(b"#"))
        if os.path.exists(ref_bin):
            walls = []
            for _ in range(2):
                t = time.perf_counter()
                q = subprocess.run([ref_bin] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                walls.append(time.perf_counter() - t)
            out["reference_wall_s"] = min(walls)
            out["reference_reads_per_s"] = rd.n / out["reference_wall_s"]
            out["vcf_identical_to_reference"] = bool(q.returncode == 0 and q.stdout == p.stdout)
            # the reference's only parallel mode: one process per -c region (src/indelminer.c:536-542,711-713), 8 at a time
            nproc = 8
            clen = len(refs[0])
            regs = ["%s:%d-%d" % (contigs[0][0], 1 + k * clen // nproc, (k + 1) * clen // nproc) for k in range(nproc)]
            t = time.perf_counter()
            procs = [subprocess.Popen([ref_bin, "-c", rg] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for rg in regs]
            outs = [pr.communicate()[0] for pr in procs]
            out["reference_8proc_wall_s"] = time.perf_counter() - t
            out["reference_8proc_reads_per_s"] = rd.n / out["reference_8proc_wall_s"]
            body = lambda b: b"".join(l + b"\n" for l in b.splitlines() if not l.startswith(b"#"))
            out["reference_8proc_same_records_as_1proc"] = bool(b"".join(body(o) for o in outs) == body(q.stdout))
        out["product_reads_per_s"] = rd.n / out["product_wall_s"]
        out["note"] = "whole programs incl. process start, BAM decode, GPU context creation and reference upload; product: one contig = one walker thread with 4 BGZF inflate workers, a GPU start-up helper thread (the GPU context creation, 0.1-0.25 s by run, is most of it on this input); reference: 1 thread, and its one-process-per-region mode at 8 processes; each program twice, the faster run counted"
        return out


def end_to_end_config3():
    """BASELINE configs[2] AT FULL SIZE (8 contigs x 6.25 Mb, 30x: 15 M reads, a 430 MB BAM) through the product CLI on this one GPU,
    no config file; the VCF's md5 must be the one the compiled reference printed for this input (tests/golden/large_config3.json,
    508 s of reference time in the build container).  A whole-program number at a size where start-up is not the run.  Beside it the
    CPU side on this box's cores: tests/shim/indelminer_shim -- this host driver over the CPU oracle, record at a time, i.e. the
    reference's path without its per-candidate strlen of the contig -- as one process and as eight -c processes at once (the
    reference's only parallel mode, src/indelminer.c:536-542)."""
    import hashlib
    import importlib.util
    import subprocess
    import tempfile
    from indelminer_amd import build
    spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(ROOT, "tests", "golden", "make_golden_large.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "large_config3.json")))
    prod = build.HOST_BIN
    with tempfile.TemporaryDirectory() as td:
        t = time.perf_counter()
        n, flags = mg.materialise("config3", td)
        out = {"workload": "BASELINE configs[2] at full size: 8 contigs x 6.25 Mb, 100 bp PE at 30x, every seventh planted event a 150-900 bp deletion; "
                           "no config file (insert lengths estimated by the run)", "reads": int(n), "bam_bytes": os.path.getsize(td + "/aln.bam"),
               "generation_s": time.perf_counter() - t, "reference_wall_s_build_container": want.get("reference_wall_s")}
        cmd = flags + ["ref.fa", "s=aln.bam"]
        walls = []
        for _ in range(3):
            t = time.perf_counter()
            p = subprocess.run([prod] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            walls.append(time.perf_counter() - t)
        d = mg.digest(p.stdout)
        out.update(product_rc=p.returncode, product_wall_s_runs=walls, product_wall_s=float(np.median(walls)), product_reads_per_s=n / float(np.median(walls)),
                   vcf_records=d["records"], product_md5=d["md5"], product_md5_is_the_references=bool(p.returncode == 0 and d["md5"] == want["md5"]))
        try:
            from tests.support.shimbuild import build_shim
            shim = build_shim()
            env = dict(os.environ, INDELMINER_PIPELINE="host")
            t = time.perf_counter()
            q = subprocess.run([shim] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
            w1 = time.perf_counter() - t
            out["cpu_1_process"] = {"kind": "port", "wall_s": w1, "reads_per_s": n / w1, "cores": 1, "md5_is_the_references": hashlib.md5(q.stdout).hexdigest() == want["md5"]}
            t = time.perf_counter()
            procs = [subprocess.Popen([shim, "-c", "ctg%d" % i] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for i in range(8)]
            outs = [pr.communicate()[0] for pr in procs]
            w8 = time.perf_counter() - t
            body = lambda b: [l for l in b.splitlines() if not l.startswith(b"#")]
            out["cpu_8_processes"] = {"kind": "port", "wall_s": w8, "reads_per_s": n / w8, "cores": 8,
                                      "same_records_as_the_whole_run": sum((body(o) for o in outs), []) == body(q.stdout)}
            out["product_over_cpu_1_process"] = w1 / out["product_wall_s"]
            out["product_over_cpu_8_processes"] = w8 / out["product_wall_s"]
        except Exception as ex:
            out["cpu_error"] = str(ex)
        return out


def end_to_end_config5():
    """BASELINE configs[4] at the size of configs[2]: tumour + normal, 8 contigs x 6.25 Mb at 30x each (15 M reads per sample; the tumour
    is the normal's genome with somatic indels on top).  The product CLI runs discovery on the tumour, then annotate mode
    (-q 0 -a -e 1, README.md:116) on the normal with the tumour's VCF; both outputs must carry the md5 the compiled reference
    printed for these inputs (tests/golden/large_config5.json: 451 s + 403 s of reference time in the build container).
    Throughput of the annotate step = the normal's reads / its wall time."""
    import importlib.util
    import re
    import subprocess
    import tempfile
    from indelminer_amd import build
    spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(ROOT, "tests", "golden", "make_golden_large.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "large_config5.json")))
    prod = build.HOST_BIN
    with tempfile.TemporaryDirectory() as td:
        t = time.perf_counter()
        n_t, n_n = mg.materialise_tn(td)
        out = {"workload": "BASELINE configs[4]: tumour + normal, 8 contigs x 6.25 Mb, 100 bp PE at 30x each; discovery on the tumour, then -q 0 -a -e 1 on the normal",
               "tumor_reads": n_t, "normal_reads": n_n, "generation_s": time.perf_counter() - t,
               "reference_wall_s_build_container": {"discovery": want["reference_discovery_wall_s"], "annotate": want["reference_annotate_wall_s"]}}
        env = dict(os.environ, INDELMINER_TIMING="1")
        t = time.perf_counter()
        p = subprocess.run([prod] + mg.TN_DISCOVER, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        w_disc = time.perf_counter() - t
        open(td + "/tumor.vcf", "wb").write(p.stdout)
        t = time.perf_counter()
        a = subprocess.run([prod] + mg.TN_ANNOTATE, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        w_ann = time.perf_counter() - t
        d1, d2 = mg.digest(p.stdout), mg.digest(a.stdout)
        body = [l for l in a.stdout.splitlines() if not l.startswith(b"#")]
        sw = re.search(r"annotate mode: (\d+) im_support_batch calls, (\d+) tasks, (\d+) cells, ([0-9.]+) s", a.stderr.decode(errors="replace"))
        out.update(discovery={"rc": p.returncode, "wall_s": w_disc, "reads_per_s": n_t / w_disc, "vcf_records": d1["records"], "md5": d1["md5"],
                              "md5_is_the_references": bool(p.returncode == 0 and d1["md5"] == want["tumor"]["md5"])},
                   annotate={"rc": a.returncode, "wall_s": w_ann, "reads_per_s": n_n / w_ann, "vcf_records": d2["records"], "md5": d2["md5"],
                             "tagged_normal": sum(1 for l in body if l.endswith(b";normal")),
                             "md5_is_the_references": bool(a.returncode == 0 and d2["md5"] == want["annotate"]["md5"]),
                             "support_kernel": ({"calls": int(sw.group(1)), "tasks": int(sw.group(2)), "cells": int(sw.group(3)), "s_in_calls": float(sw.group(4)),
                                                 "note": "affine Smith-Waterman of annotate mode (src/variant.c:1246-1424); only variants the normal's own discovery "
                                                         "does not already carry reach it"} if sw else None)})
        out["product_md5s_are_the_references"] = bool(out["discovery"]["md5_is_the_references"] and out["annotate"]["md5_is_the_references"])
        out["speedup_over_reference_build_container"] = {"discovery": want["reference_discovery_wall_s"] / w_disc, "annotate": want["reference_annotate_wall_s"] / w_ann}
        return out


def _mg_product_run(rank, world, local_rank, dist, td, flags, n_reads, piece_bytes, single_wall=None, single_vcf=None):
    """the product CLI on every rank of the job over the input in td (ref.fa, aln.bam); rank 0 also runs it as ONE process and compares
    the bytes.  Returns (on rank 0) wall time = max over ranks, reads/s, per-rank phase times and what RCCL said about the communicator."""
    import re
    import subprocess
    from indelminer_amd import build
    cmd = [build.HOST_BIN] + flags + ["ref.fa", "s=aln.bam"]
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), INDELMINER_RENDEZVOUS=td + "/rdv", INDELMINER_RUN_TOKEN=td,
               INDELMINER_TIMING="1", INDELMINER_MG_TIMEOUT="120")          # a rank that waits longer than that for the others gives up (default 600 s)
    if piece_bytes:
        env["INDELMINER_PIECE_BYTES"] = str(piece_bytes)
    if world == 1:
        env["INDELMINER_FORCE_MGPU"] = "1"
    if os.environ.get("IM_BENCH_ONE_DEVICE") == "1":
        env["INDELMINER_DEVICE"] = "0"
    t = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=400)
        rc, vcf, err = p.returncode, p.stdout, p.stderr.decode(errors="replace")
    except Exception as ex:
        rc, vcf, err = -1, b"", str(ex)
    wall = time.perf_counter() - t
    phases = {}
    for m_ in re.finditer(r"\[timing\] (.+?)\s+([0-9.]+) ms", err):
        phases[m_.group(1).strip()] = phases.get(m_.group(1).strip(), 0.0) + float(m_.group(2))
    keep = ("the walk of this rank's pieces", "the ranks' walk logs exchanged", "walked groups exchanged between the ranks", "depth arrays summed over the ranks",
            "GPU context + reference upload", "read FASTA", "device: realign + flush cuts + group-by", "results to the host", "replay workers drained")
    mine = {"rank": rank, "wall_s": wall, "rc": rc, "rccl": re.findall(r"RCCL communicator of (\d+) ranks", err)[:1],
            "phases_ms": {k: round(sum(v for n_, v in phases.items() if n_.startswith(k)), 1) for k in keep}}
    per_rank = [mine]
    ok = rc == 0
    if dist is not None:
        import torch
        w = torch.tensor([wall], dtype=torch.float64); dist.all_reduce(w, op=dist.ReduceOp.MAX); wall = float(w[0])
        k = torch.tensor([1 if ok else 0]); dist.all_reduce(k, op=dist.ReduceOp.MIN); ok = bool(int(k[0]))
        got = [None] * world if rank == 0 else None
        dist.gather_object(mine, got, dst=0)
        if rank == 0:
            per_rank = got
    if rank != 0:
        return None
    out = {"n_gpus": world, "reads": n_reads, "all_ranks_ok": ok, "wall_s": wall, "reads_per_s": n_reads / wall if ok else None,
           "rccl_ranks_seen": sorted({int(x) for r_ in per_rank for x in r_["rccl"]}), "per_rank": per_rank}
    if not ok:
        out["rank0_stderr_tail"] = err[-400:]
    if single_vcf is None:
        env1 = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
        if os.environ.get("IM_BENCH_ONE_DEVICE") == "1":
            env1["INDELMINER_DEVICE"] = "0"
        t = time.perf_counter()
        q = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1, timeout=400)
        single_wall, single_vcf = time.perf_counter() - t, (q.stdout if q.returncode == 0 else None)
    out["single_process_wall_s"] = single_wall
    out["vcf_records"] = sum(1 for l in (single_vcf or b"").splitlines() if not l.startswith(b"#"))
    out["vcf_identical_to_single_process"] = bool(ok and single_vcf is not None and single_vcf == vcf)
    out["speedup_over_single_process"] = (single_wall / wall) if ok else None
    return out


def end_to_end_multi(rank, world, local_rank, dist):
    """STRONG scaling of the PRODUCT on the job's GPUs: every rank starts `indelminer` as a child with its own
    RANK / LOCAL_RANK / WORLD_SIZE on the SAME input -- contigs owned longest-first, pieces walked by the least-loaded rank, ONE RCCL
    all-gather of the ranks' walk logs, walked groups to their owners in ONE RCCL send / receive group, one sum of the depth arrays,
    rank 0 prints the VCF -- and rank 0 compares that VCF byte for byte with the single-process run.  Two inputs:
      config3     BASELINE configs[2] at full size (8 contigs x 6.25 Mb, 15 M reads, no config file): contigs over the ranks
      one_contig  BASELINE configs[1] (ONE 1 Mb contig) cut into about 3 x world pieces: every rank walks pieces of the one contig
    Reported beside the kernel-level numbers, never `value`."""
    import importlib.util
    import shutil
    import tempfile
    from indelminer_amd import bamwrite, build
    if not os.path.exists(build.HOST_BIN):
        return None
    out = {"what": "the product CLI, one process per GPU, the same input for every N (strong scaling); per rank the [timing] phases it printed"}
    strong = {}
    # ---- one contig in pieces
    box = [None, 0, 0]
    if rank == 0:
        box[0] = tempfile.mkdtemp(prefix="im_mgpu_")
        refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30, read_len=100)
        contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
        bamwrite.write_fasta(box[0] + "/ref.fa", contigs, refs)
        rawrec.write_bam_fast(box[0] + "/aln.bam", contigs, rd, level=6)
        open(box[0] + "/cfg.txt", "w").write("IL generic 300 %d\n" % rd.range_max)
        box[1] = int(rd.n)
        box[2] = os.path.getsize(box[0] + "/aln.bam")
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    td, n_reads, bam_bytes = box
    try:
        strong["one_contig"] = _mg_product_run(rank, world, local_rank, dist, td, ["-i", "cfg.txt"], n_reads, max(bam_bytes // (3 * max(world, 1)), 200000))
    except Exception as ex:
        strong["one_contig"] = {"error": str(ex)}
    if dist is not None:
        dist.barrier()
    if rank == 0:
        shutil.rmtree(td, ignore_errors=True)
    # ---- configs[2] at full size
    if os.environ.get("IM_BENCH_MG_CONFIG3", "1") == "1":
        box = [None, 0]
        if rank == 0:
            spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(ROOT, "tests", "golden", "make_golden_large.py"))
            mg = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mg)
            box[0] = tempfile.mkdtemp(prefix="im_mgpu3_")
            box[1], _ = mg.materialise("config3", box[0])
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        td, n_reads = box
        try:
            r3 = _mg_product_run(rank, world, local_rank, dist, td, [], int(n_reads), 0)
            if rank == 0:
                import hashlib
                want = json.load(open(os.path.join(ROOT, "tests", "golden", "large_config3.json")))
                r3["workload"] = "BASELINE configs[2] at full size, no config file"
                r3["vcf_records_of_the_reference"] = want["records"]
            strong["config3"] = r3
        except Exception as ex:
            strong["config3"] = {"error": str(ex)}
        if dist is not None:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(td, ignore_errors=True)
    if rank != 0:
        return None
    out["strong"] = strong
    good = [v for v in strong.values() if isinstance(v, dict) and "error" not in v]
    out["all_ranks_ok"] = bool(good) and len(good) == len(strong) and all(v["all_ranks_ok"] for v in good)
    out["vcf_identical_to_single_process"] = bool(good) and len(good) == len(strong) and all(v["vcf_identical_to_single_process"] for v in good)
    return out


def cpu_port_baseline(ref, cand, read_len, n_reads_total, m):
    """The hoisted CPU baseline: the oracle restatement takes the contig length as an
    argument, i.e. the reference's path WITHOUT its per-candidate strlen of the contig (src/alignment.c:771).
    Same sample as the reference leg, one thread."""
    from tests.support import oraclebind as ob
    P = ob.params()
    contig = ref.tobytes()
    t = time.perf_counter()
    for i in range(m):
        ob.realign(P, contig, len(contig), int(cand["anchor"][i]), int(cand["range_max"][i]), bytes(cand["bases"][i]))
    dt = time.perf_counter() - t
    frac = len(cand["index"]) / float(n_reads_total)
    return {"value": (m / frac) / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "%d candidate reads through oracle/im_oracle.c (no contig strlen), %.2f s, 1 thread" % (m, dt),
            "candidates_per_s": m / dt}


def long_reads_measure(device, reps=12):
    """A 2 x 300 library (1 Mb contig at 30x, insert ~ N(900, 50)): the candidates of its delivered reads through the realign launches --
    realign_kernel turns reads beyond 255 bases away, realign_long_kernel (sixteen read positions per lane) takes them -- timed with
    HIP events on a resident batch, every result checked against the CPU oracle on a sample.  Algorithmic bytes as for the short
    kernel, from the window / piece sizes the kernel records."""
    from tests.support import gpucmp, oraclebind as ob
    L = 300
    refs, rd = synth.simulate(seed=8, ref_len=1_000_000, coverage=30, read_len=L, isize_mean=900, isize_sd=50, isize_min=700, isize_max=1100)
    cand = synth.candidates(rd)
    n = len(cand["index"])
    ctx = capi.Context(device)
    try:
        ctx.set_reference([refs[0].tobytes()])
        ctx.expect_read_length(L)
        stride = (L + 3) // 4 * 4
        bases = np.zeros((n, stride), dtype=np.uint8)
        bases[:, :L] = cand["bases"]
        flat = np.concatenate([bases.reshape(-1), np.zeros(16, np.uint8)])
        d_bases = capi.DevBuf(ctx, flat.nbytes).upload(flat)
        d_off = capi.DevBuf(ctx, 8 * n).upload(np.arange(n, dtype=np.int64) * stride)
        d_len = capi.DevBuf(ctx, 4 * n).upload(np.full(n, L, np.int32))
        d_tid = capi.DevBuf(ctx, 4 * n).upload(np.zeros(n, np.int32))
        d_anchor = capi.DevBuf(ctx, 4 * n).upload(cand["anchor"].astype(np.int32))
        d_range = capi.DevBuf(ctx, 4 * n).upload(cand["range_max"].astype(np.int32))
        d_res = capi.DevBuf(ctx, 512 * n)
        batch = capi.DevBatch(n, d_bases.ptr, d_off.ptr, d_len.ptr, d_tid.ptr, d_anchor.ptr, d_range.ptr, d_res.ptr, None, None, None)
        P = capi.params()
        L_ = capi.lib()
        tm = capi.Timer(ctx)
        ts = []
        for _ in range(reps + 2):
            tm.start(ctx.stream)
            ctx._check(L_.im_dev_realign(ctx.h, C.byref(P), C.byref(batch), ctx.stream))
            tm.stop(ctx.stream)
            ts.append(tm.elapsed_ms())
        ms = float(np.median(ts[2:]))
        res = d_res.download(capi.RESULT_DTYPE, n)
        alg = algorithmic_bytes(res)
        contig = refs[0].tobytes()
        Po = ob.params()
        bad = 0
        m = min(n, 1500)
        for j in range(m):
            st, r_ = ob.realign(Po, contig, len(contig), int(cand["anchor"][j]), int(cand["range_max"][j]), bytes(cand["bases"][j]))
            bad += gpucmp.hip_vs_oracle(res[j], st, r_) is not None
        # the same batch with -g 2: reads beyond 255 bases with a band take the general pass (im_realign_any.hip)
        Pg = capi.params(numgaps=2)
        tg = []
        for _ in range(4):
            tm.start(ctx.stream)
            ctx._check(L_.im_dev_realign(ctx.h, C.byref(Pg), C.byref(batch), ctx.stream))
            tm.stop(ctx.stream)
            tg.append(tm.elapsed_ms())
        msg_ = float(np.median(tg[1:]))
        resg = d_res.download(capi.RESULT_DTYPE, n)
        Pog = ob.params(numgaps=2)
        badg = 0
        mg = min(n, 300)
        for j in range(mg):
            st, r_ = ob.realign(Pog, contig, len(contig), int(cand["anchor"][j]), int(cand["range_max"][j]), bytes(cand["bases"][j]))
            badg += gpucmp.hip_vs_oracle(resg[j], st, r_) is not None
        gapped = {"numgaps": 2, "kernel": "realign_any_kernel (general pass: band searches by the whole wave read by read, dynamic programs one lane per read)", "ms": msg_,
                  "candidates_per_s": n / (msg_ * 1e-3), "evidence_found": int((resg["status"] == 1).sum()),
                  "identical_to_the_oracle_on_the_sample": bool(badg == 0), "sample": mg}
        return {"workload": "2 x 300 library: 1 Mb contig, 30x, insert ~ N(900, 50); the candidate reads of its %d delivered reads" % rd.n,
                "with_gaps": gapped,
                "candidates": n, "evidence_found": int((res["status"] == 1).sum()), "ms_both_launches": ms, "candidates_per_s": n / (ms * 1e-3),
                "delivered_reads_per_s_equivalent": rd.n / (ms * 1e-3), "algorithmic_bytes": alg, "achieved_gbs": alg / (ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "identical_to_the_oracle_on_the_sample": bool(bad == 0), "sample": m}
    finally:
        ctx.close()


def triage_with_aligner_tags(device, rd, contig, reps=8):
    """The triage launches alone on the step's records WITH what an aligner writes into the aux area (NM:C MD:Z AS:C XS:C MC:Z in front
    of MQ:C and an RG:Z tag, 46 bytes; the synthetic records of the timed step carry MQ:C alone): the tag walk of fetch_func's RG / MQ
    look-up is then most of classify.  Candidate count and classes must be what they are without the tags."""
    ctx = capi.Context(device)
    try:
        ctx.set_reference([contig])
        ctx.set_insert_ranges(["generic"], [rd.range_max])
        out = {}
        cls = {}
        L_ = capi.lib()
        for name, kw in (("plain", {}), ("tagged", dict(rg="generic", aux_prefix=b"NMC\x00" + b"MDZ100\x00" + b"ASC\x64" + b"XSC\x00" + b"MCZ100M\x00"))):
            raw, off = rawrec.records(rd, qual=False, **kw)
            pipe = capi.Pipeline(ctx, rd.n, len(raw), cap_cand=max(4096, rd.n // 8), read_len_max=rd.read_len)
            pipe.upload(raw, off)
            tp = capi.TriageParams(pipe.tp.qthreshold, pipe.tp.ethreshold_vcfcheck, pipe.tp.maxpedelsize, 0, 0, 1)
            tm = capi.Timer(ctx)
            ts = []
            for _ in range(reps):
                ctx._check(L_.im_stream_sync(ctx.h, ctx.stream))
                tm.start(ctx.stream)
                ctx._check(L_.im_dev_triage(ctx.h, C.byref(tp), C.byref(pipe.recs), C.byref(pipe.cands), pipe.d_ts.ptr, pipe.ts_bytes, ctx.stream))
                tm.stop(ctx.stream)
                ts.append(tm.elapsed_ms())
            out[name] = {"record_bytes": int(len(raw)), "ms_3_launches": float(np.median(ts[2:]))}
            cls[name] = (pipe.d_class.download(np.uint8, rd.n).tobytes(), int(pipe.fetch_counts()[0]))
            del pipe
        out["same_classes_and_candidates"] = bool(cls["plain"] == cls["tagged"])
        return out
    finally:
        ctx.close()


def other_parameters_measure(device, reps=10):
    """The flags users run besides the defaults, on the configs[1] candidates (12 232 reads of 100 bases): -g 1 / 2 / 5 / 12
    (realign_band_kernel: a lane per band diagonal), -k 8 / 13 (the table by the k-mer's first six bases), -k 14 (the hash).  Each
    setting: the realign launch alone on the device (HIP events), a sample of its results against the CPU oracle."""
    from tests.support import gpucmp, oraclebind as ob
    L = 100
    refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30, read_len=L)
    cand = synth.candidates(rd)
    n = len(cand["index"])
    ctx = capi.Context(device)
    try:
        ctx.set_reference([refs[0].tobytes()])
        stride = (L + 3) // 4 * 4
        bases = np.zeros((n, stride), dtype=np.uint8)
        bases[:, :L] = cand["bases"]
        flat = np.concatenate([bases.reshape(-1), np.zeros(16, np.uint8)])
        d_bases = capi.DevBuf(ctx, flat.nbytes).upload(flat)
        d_off = capi.DevBuf(ctx, 8 * n).upload(np.arange(n, dtype=np.int64) * stride)
        d_len = capi.DevBuf(ctx, 4 * n).upload(np.full(n, L, np.int32))
        d_tid = capi.DevBuf(ctx, 4 * n).upload(np.zeros(n, np.int32))
        d_anchor = capi.DevBuf(ctx, 4 * n).upload(cand["anchor"].astype(np.int32))
        d_range = capi.DevBuf(ctx, 4 * n).upload(cand["range_max"].astype(np.int32))
        d_res = capi.DevBuf(ctx, 512 * n)
        batch = capi.DevBatch(n, d_bases.ptr, d_off.ptr, d_len.ptr, d_tid.ptr, d_anchor.ptr, d_range.ptr, d_res.ptr, None, None, None)
        L_ = capi.lib()
        tm = capi.Timer(ctx)
        contig = refs[0].tobytes()
        out = {"workload": "the %d candidate reads of the configs[1] step (100 bases), one realign launch per setting" % n, "settings": []}
        ok = True
        for kw in (dict(numgaps=1), dict(numgaps=2), dict(numgaps=5), dict(numgaps=12), dict(klength=8), dict(klength=13), dict(klength=14)):
            P = capi.params(**kw)
            ts = []
            for _ in range(reps + 2):
                tm.start(ctx.stream)
                ctx._check(L_.im_dev_realign(ctx.h, C.byref(P), C.byref(batch), ctx.stream))
                tm.stop(ctx.stream)
                ts.append(tm.elapsed_ms())
            ms = float(np.median(ts[2:]))
            res = d_res.download(capi.RESULT_DTYPE, n)
            Po = ob.params(**kw)
            m = min(n, 250)
            bad = 0
            for j in range(m):
                st, r_ = ob.realign(Po, contig, len(contig), int(cand["anchor"][j]), int(cand["range_max"][j]), bytes(cand["bases"][j]))
                bad += gpucmp.hip_vs_oracle(res[j], st, r_) is not None
            ok = ok and bad == 0
            out["settings"].append({"flags": " ".join("-%s %d" % ("g" if k_ == "numgaps" else "k", v_) for k_, v_ in kw.items()),
                                    "ms": ms, "candidates_per_s": n / (ms * 1e-3), "evidence_found": int((res["status"] == 1).sum()),
                                    "identical_to_the_oracle_on_the_sample": bool(bad == 0), "sample": m})
        out["identical_to_the_oracle_on_the_samples"] = bool(ok)
        return out
    finally:
        ctx.close()


def shard3_measure(device, steps=24, warmup=4):
    """The same device pass on ONE GPU's share of BASELINE configs[2] (a 6.25 Mb contig at 30x, every seventh planted event a
    150-900 bp deletion: 1.9 M delivered reads, ~76 k candidates, ~19 READCHUNK flushes per step): the launch sizes a real
    per-GPU shard gives, beside the configs[1] step `value` is quoted on."""
    L = 100
    refs, rd = synth.simulate(seed=2, ref_len=6_250_000, coverage=30, read_len=L, big_every=7)
    cand = synth.candidates(rd)
    ctx = capi.Context(device)
    try:
        ctx.set_reference([refs[0].tobytes()] * PIPELINE_DEPTH)
        ctx.set_insert_ranges(["generic"], [rd.range_max])
        ps = PipeStep(ctx, rd, PIPELINE_DEPTH)
        ps.step(); ps.sync()
        c0, _, _ = ps.results()
        assert int(c0[0]) == len(cand["index"]) and int(c0[3]) == 0 and int(c0[4]) == 0, c0
        ps.capture()
        for _ in range(warmup):
            ps.step()
        ps.sync()
        p0 = ps.sets[0]["pipe"]
        timers = [(capi.Timer(ctx), p0.realign_call_index if (i // 4) % 2 == 0 else p0.triage_call_index) if i % 4 == 0 else (None, None)
                  for i in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ps.step(timers[i][0], timers[i][1])
        ps.sync()
        elapsed = time.perf_counter() - t0
        realign_ms = np.array([tm.elapsed_ms() for tm, at in timers if tm is not None and at == p0.realign_call_index])
        triage_ms = np.array([tm.elapsed_ms() for tm, at in timers if tm is not None and at == p0.triage_call_index])
        cnt, res, counts = ps.results()
        alg = algorithmic_bytes(res)
        n_band = int(res["n_band"].sum())
        tri_bytes = ps.record_bytes + 4 * (rd.n + 1) + rd.n + len(cand["index"]) * (((L + 3) // 4) * 4 + 24 + 48)
        q_realign, q_triage = ps.quiet_launch_times(8)
        ach = alg / (q_realign * 1e-3) / 1e9
        return {"workload": "one GPU's share of BASELINE configs[2]: 6.25 Mb contig, 30x, big_every=7",
                "reads_per_step": int(rd.n), "candidates_per_step": int(len(cand["index"])), "flushes_per_step": len(ps.flushes),
                "evidence_nodes_per_step": int(counts[1]), "clusters_per_step": int(counts[0]),
                "steps": steps, "ms_per_step": elapsed / steps * 1e3, "reads_per_s": rd.n * steps / elapsed,
                "candidates_per_s": len(cand["index"]) * steps / elapsed,
                "band_alignments_per_s": n_band * steps / elapsed, "gcups": 2.0 * L * n_band * steps / elapsed / 1e9,
                "realign": {"avg_launch_ms": q_realign, "avg_launch_ms_inside_the_overlapped_steps": float(realign_ms.mean()) if len(realign_ms) else None,
                            "algorithmic_bytes_per_launch": alg, "achieved_gbs": ach,
                            "frac_of_hbm_peak": ach / HBM_PEAK_GBS,
                            "occupancy_rounds": len(cand["index"]) / (256.0 * 24)},
                "triage": {"avg_ms_3_launches": q_triage, "avg_ms_inside_the_overlapped_steps": float(triage_ms.mean()) if len(triage_ms) else None,
                           "algorithmic_bytes_per_launch": int(tri_bytes), "achieved_gbs": tri_bytes / (q_triage * 1e-3) / 1e9}}
    finally:
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)       # ~150 ms of timed region at ~75 us per step
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ref-len", type=int, default=1_000_000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--big-every", type=int, default=0, help="every k-th planted event a 150-900 bp deletion (config-3 style shards)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard3", action="store_true", help="skip the config-3 per-GPU shard measurement (N=1 only)")
    ap.add_argument("--no-config3", action="store_true", help="skip the whole-program leg on BASELINE configs[2] at full size (N=1 only, about a minute)")
    ap.add_argument("--no-config5", action="store_true", help="skip the tumour / normal leg (BASELINE configs[4]: discovery + annotate mode, N=1 only, about a minute and a half)")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: gloo ("[Gloo] Rank 0 is connected ...") and librccl (its
    # version banner) both print on file descriptor 1, so the descriptor points at stderr for the whole run
    # and the line is written to the saved original at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    failed = False
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist     # control plane only (gloo): barrier + max over ranks
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    # ---- data: one contig per rank, seeded by rank (weak scaling) ----
    L = 100
    refs, rd = synth.simulate(seed=1 + rank, ref_len=args.ref_len, coverage=args.coverage, read_len=L, big_every=args.big_every)
    cand = synth.candidates(rd)
    n_reads = rd.n
    n_cand = len(cand["index"])

    # IM_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank control flow on a one-GPU box
    ctx = capi.Context(0 if os.environ.get("IM_BENCH_ONE_DEVICE") == "1" else local_rank)
    ctx.set_reference([refs[0].tobytes()] * PIPELINE_DEPTH)        # buffer set k works on contig k (PipeStep)
    ctx.set_insert_ranges(["generic"], [rd.range_max])
    ps = PipeStep(ctx, rd, PIPELINE_DEPTH)

    # ---- no data-path collective in the timed step: contigs (and pieces of contigs) are independent on the device; the product's
    # exchanges -- one all-gather of the ranks' logs before the walk, one sum of the depth arrays when pieces of a contig were
    # walked by several ranks -- happen once per RUN and are measured where they happen, in end_to_end_multi_gpu ----
    ps.step(); ps.sync()
    c0, res0, counts0 = ps.results()
    assert int(c0[0]) == n_cand and int(c0[3]) == 0 and int(c0[4]) == 0, (c0, n_cand)
    digest0 = ps.digest(ps.last)           # the pass on its own, nothing else on the device
    collective = "none in the timed step (shards are independent); the product's per-run exchanges are in end_to_end_multi_gpu"

    def barrier():
        ps.sync()
        if dist is not None:
            dist.barrier()

    if os.environ.get("IM_BENCH_GRAPH", "1") == "1":
        ps.capture()
    for _ in range(args.warmup):
        ps.step()
    barrier()

    # ---- the timed region: EXACTLY --steps steps between barrier + device sync on both sides, max over ranks.  A region of 20 steps
    # is 1.3 ms, so the region is repeated (same bracket every time) until MIN_TIMED_S have been measured; the line carries the
    # MEDIAN region and the spread.  Nothing but graph launches is issued inside a region.
    def region():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ps.step()
        ps.sync()
        t1 = time.perf_counter()
        if dist is not None:
            dist.barrier()
        e = t1 - t0
        if dist is not None:
            t = torch.tensor([e], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t[0])
        return e

    regions = [region()]
    n_regions = int(min(MAX_REGIONS, max(MIN_REGIONS, np.ceil(MIN_TIMED_S / max(regions[0], 1e-6)))))
    if dist is not None:
        t = torch.tensor([n_regions], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_regions = int(t[0])
    for _ in range(n_regions - 1):
        regions.append(region())
    regions = np.array(regions)
    elapsed = float(np.median(regions))
    if dist is not None:
        tot = torch.tensor([n_reads, n_cand], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_reads, total_cand = int(tot[0]), int(tot[1])
    else:
        total_reads, total_cand = n_reads, n_cand

    # ---- outside the timing: overlapped steps with ONE launch of every fourth step bracketed by HIP events on its stream
    timers = []
    p0 = ps.sets[0]["pipe"]
    for i in range(BRACKET_STEPS):
        timers.append((capi.Timer(ctx), p0.realign_call_index if (i // 4) % 2 == 0 else p0.triage_call_index) if i % 4 == 2 else (None, None))
    for i in range(BRACKET_STEPS):
        ps.step(timers[i][0], timers[i][1])
    ps.sync()

    # every buffer set must hold exactly what the pass leaves when it runs alone: the timed passes overlap on PIPELINE_DEPTH streams
    digests = [ps.digest(st_) for st_ in ps.sets]
    overlap_ok = all(dg == digest0 for dg in digests)
    try:
        depth_ok = bool(ps.depth_check(rd, len(refs[0])))
    except Exception as ex:
        depth_ok = "not checked: %s" % ex
    realign_ms = np.array([tm.elapsed_ms() for tm, at in timers if tm is not None and at == p0.realign_call_index])
    triage_ms = np.array([tm.elapsed_ms() for tm, at in timers if tm is not None and at == p0.triage_call_index])
    cnt, res, counts = ps.results()
    ncl, nodes = int(counts[0]), int(counts[1])
    alg_bytes = algorithmic_bytes(res)
    q_realign, q_triage = ps.quiet_launch_times()
    kern_s = q_realign * 1e-3
    achieved = alg_bytes / kern_s / 1e9
    n_band = int(res["n_band"].sum())
    # triage: every record byte in, per candidate the padded read + 24 B of scalars + 48 B of slots out, 1 B class per record
    tri_bytes = ps.record_bytes + 4 * (n_reads + 1) + n_reads + n_cand * (((L + 3) // 4) * 4 + 24 + 48)

    e2e_mg = None
    if world > 1 or os.environ.get("IM_BENCH_FORCE_COMM") == "1":
        try:
            e2e_mg = end_to_end_multi(rank, world, local_rank, dist)
        except Exception as ex:                 # plumbing; never hides the kernel numbers
            e2e_mg = {"error": str(ex)}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed
        cpu = None
        cpu_port = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, ref_out = cpu_reference_baseline(refs[0], cand, L, n_reads)
            if ref_out is not None:
                m = len(ref_out)
                st = res["status"][:m]
                ok = (np.where(st == 1, res["n_ev"][:m], 0) == ref_out[:, 0])
                has = ref_out[:, 0] > 0
                ev0 = res["ev"][:m, 0]
                ok &= ~has | ((ev0["cls"] == ref_out[:, 1]) & (ev0["b1"] == ref_out[:, 2]) & (ev0["b2"] == ref_out[:, 3]))
                parity = "identical to the reference on %d sampled reads" % m if bool(ok.all()) else \
                    "MISMATCH on %d of %d sampled reads" % (int((~ok).sum()), m)
                try:
                    cpu_port = cpu_port_baseline(refs[0], cand, L, n_reads, min(m, 20000))
                except Exception as ex:
                    cpu_port = {"error": str(ex)}
        e2e = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                e2e = end_to_end(refs, rd)
            except Exception as ex:              # plumbing; never hides the kernel numbers
                e2e = {"error": str(ex)}
        try:
            peak_measured = copy_peak_gbs(ctx)
        except Exception:
            peak_measured = None
        try:
            h2d = h2d_pinned_gbs(ctx)
        except Exception:
            h2d = None
        try:
            s_pcie = ps.measure_with_uploads(min(max(args.steps, 50), 400))
        except Exception as ex:
            s_pcie = None
            sys.stderr.write("measure_with_uploads: %s\n" % ex)
        line = {
            "metric": "reads/sec through split-read realign+cluster; VCF diff-clean vs reference",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_is": "the median of %d timed regions of --steps steps each" % len(regions),
            "timed_regions": {"n": int(len(regions)), "timed_s_in_total": float(regions.sum()),
                              "ms_per_step_min": float(regions.min() / args.steps * 1e3), "ms_per_step_max": float(regions.max() / args.steps * 1e3),
                              "spread_iqr_over_median": float((np.percentile(regions, 75) - np.percentile(regions, 25)) / elapsed),
                              "spread_range_over_median": float((regions.max() - regions.min()) / elapsed)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic %.1f Mb contig per GPU, 100 bp PE reads at %gx, seeded "
                                   "1-50 bp indels every ~2 kb, BWA-like S/I/D emission; -k 6 -g 0 -s 1000 -n 10 -q 10"
                                   % (args.ref_len / 1e6, args.coverage),
                       "reads_per_step": total_reads, "candidates_per_step": total_cand,
                       "record_bytes_per_step_rank0": ps.record_bytes,
                       "h2d_pinned_gbs": h2d,
                       "reads_per_s_with_the_records_crossing_pcie": (n_reads / s_pcie) if s_pcie else None,
                       "reads_per_s_with_the_records_crossing_pcie_note": "MEASURED: every step uploads its records from pinned host memory on its stream, "
                                                                          "then runs the pass; 4 streams overlap copies and kernels as the product's walkers do",
                       "ms_per_step_with_the_records_crossing_pcie": s_pcie * 1e3 if s_pcie else None,
                       "flushes_per_step": len(ps.flushes), "evidence_nodes_per_step_rank0": nodes, "clusters_per_step_rank0": ncl,
                       "candidates_per_s": total_cand * args.steps / elapsed,
                       "band_alignments_per_s": n_band * world * args.steps / elapsed,
                       "gcups": 2.0 * L * n_band * world * args.steps / elapsed / 1e9,
                       "pipeline_depth": PIPELINE_DEPTH, "hip_graph": os.environ.get("IM_BENCH_GRAPH", "1") == "1",
                       "parallelism": "contig-sharded x%d" % world, "collective": collective,
                       "timed_region": "the product driver's device pass over EVERY delivered record of the shard: triage "
                                       "(candidate rules, base decode, CIGAR evidence, pileup depth scatter) -> depth scan of the contig -> realign -> "
                                       "one flush cut per READCHUNK flush point -> split-read group-by; BGZF inflate and the pair table stay on the host "
                                       "(north_star) and are in end_to_end, not here",
                       "parity": parity,
                       "overlapped_passes_equal_the_pass_alone": overlap_ok,
                       "depth_range_sums_equal_the_pileup_rule": depth_ok,
                       "pass_digest": {"candidates": digest0[0], "clusters": digest0[1], "md5": digest0[2],
                                       "of_each_buffer_set_after_the_timed_region": [dg[2] for dg in digests]}},
            "roofline": {"bound": "hbm", "kernel": "realign_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic()[0],
                         "longest_stage_of_the_step": "realign_kernel" if q_realign >= q_triage else "triage (classify + emit + decode, three launches)",
                         "traffic_source": "%s: PMC passes of an EARLIER run of this command (FETCH_SIZE x 2 + WRITE_SIZE per launch), not measured in this run" % measured_traffic()[1],
                         "issue": measured_issue(),
                         "issue_note": "what binds this kernel: its reference windows come out of L2 (traffic is a fraction of the algorithmic bytes), the time goes "
                                       "into instruction issue and LDS round trips -- valu/salu/lds instructions per candidate read, VALU issue cycles over "
                                       "the SIMDs' cycles, LDS bank-conflict cycles over LDS-active cycles, from the same committed counter passes as `traffic`",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": q_realign,
                         "avg_launch_ms_note": "HIP events around the realign launch on its stream with nothing else on the device (the kernel's own "
                                               "duration, what rocprofv3 --kernel-trace reports for it); inside the timed region's overlapped steps the same "
                                               "bracket also holds the wait for a dispatch slot behind the other streams' launches",
                         "avg_launch_ms_inside_the_overlapped_steps": float(realign_ms.mean()) if len(realign_ms) else None, "launches_timed_inside": int(len(realign_ms)),
                         "min_launch_ms_inside": float(realign_ms.min()) if len(realign_ms) else None,
                         "peak_measured_copy_gbs": peak_measured,
                         "frac_of_measured_copy": (achieved / peak_measured) if peak_measured else None,
                         "triage": {"algorithmic_bytes_per_launch": int(tri_bytes), "avg_ms_3_launches": q_triage,
                                    "avg_ms_inside_the_overlapped_steps": float(triage_ms.mean()) if len(triage_ms) else None,
                                    "achieved_gbs": tri_bytes / (q_triage * 1e-3) / 1e9}},
            "cpu_baseline": cpu,
            "cpu_baseline_hoisted": cpu_port,
            "end_to_end": e2e,
            "end_to_end_multi_gpu": e2e_mg,
        }
        if world == 1 and not args.no_shard3:
            try:
                line["shard_config3"] = shard3_measure(0 if os.environ.get("IM_BENCH_ONE_DEVICE") == "1" else local_rank)
            except Exception as ex:
                line["shard_config3"] = {"error": str(ex)}
        if world == 1 and not args.no_shard3:
            try:
                line["long_reads_2x300"] = long_reads_measure(0 if os.environ.get("IM_BENCH_ONE_DEVICE") == "1" else local_rank)
            except Exception as ex:
                line["long_reads_2x300"] = {"error": str(ex)}
        if world == 1 and not args.no_shard3:
            try:
                line["other_parameters"] = other_parameters_measure(0 if os.environ.get("IM_BENCH_ONE_DEVICE") == "1" else local_rank)
            except Exception as ex:
                line["other_parameters"] = {"error": str(ex)}
            try:
                line["triage_with_aligner_tags"] = triage_with_aligner_tags(0 if os.environ.get("IM_BENCH_ONE_DEVICE") == "1" else local_rank, rd, refs[0].tobytes())
            except Exception as ex:
                line["triage_with_aligner_tags"] = {"error": str(ex)}
        if world == 1 and not args.no_cpu_baseline and not args.no_config3:
            try:
                line["end_to_end_config3"] = end_to_end_config3()
            except Exception as ex:
                line["end_to_end_config3"] = {"error": str(ex)}
        if world == 1 and not args.no_cpu_baseline and not args.no_config5:
            try:
                line["end_to_end_config5"] = end_to_end_config5()
            except Exception as ex:
                line["end_to_end_config5"] = {"error": str(ex)}
        # every self-check of the line in one place; the process exits non-zero when one of them is not true
        checks = {"candidates_counted_on_the_device_equal_the_simulator": True,      # asserted above
                  "overlapped_passes_equal_the_pass_alone": overlap_ok is True,
                  "depth_range_sums_equal_the_pileup_rule": depth_ok is True}
        if parity is not None:
            checks["realign_identical_to_the_reference_on_the_sample"] = parity.startswith("identical")
        if isinstance(e2e, dict):
            checks["end_to_end_ran"] = "error" not in e2e and e2e.get("product_rc") == 0
            if "vcf_identical_to_reference" in e2e:
                checks["end_to_end_vcf_identical_to_reference"] = e2e["vcf_identical_to_reference"] is True
        if isinstance(e2e_mg, dict):
            checks["multi_gpu_all_ranks_ok"] = e2e_mg.get("all_ranks_ok") is True
            checks["multi_gpu_vcf_identical_to_single_process"] = e2e_mg.get("vcf_identical_to_single_process") is True
            for k_, v_ in (e2e_mg.get("strong") or {}).items():
                if isinstance(v_, dict) and "vcf_identical_to_single_process" in v_:
                    checks["multi_gpu_strong_%s_vcf_identical" % k_] = v_["vcf_identical_to_single_process"] is True
        if isinstance(line.get("shard_config3"), dict):
            checks["shard_config3_ran"] = "error" not in line["shard_config3"]
        if isinstance(line.get("end_to_end_config3"), dict):
            c3 = line["end_to_end_config3"]
            checks["end_to_end_config3_md5_is_the_references"] = c3.get("product_md5_is_the_references") is True
        if isinstance(line.get("long_reads_2x300"), dict):
            checks["long_reads_identical_to_the_oracle"] = line["long_reads_2x300"].get("identical_to_the_oracle_on_the_sample") is True
            if isinstance(line.get("triage_with_aligner_tags"), dict):
                checks["aligner_tags_leave_the_classes_as_they_are"] = line["triage_with_aligner_tags"].get("same_classes_and_candidates") is True
            if isinstance(line.get("other_parameters"), dict):
                checks["other_parameters_identical_to_the_oracle"] = line["other_parameters"].get("identical_to_the_oracle_on_the_samples") is True
            if isinstance(line["long_reads_2x300"].get("with_gaps"), dict):
                checks["long_reads_with_gaps_identical_to_the_oracle"] = line["long_reads_2x300"]["with_gaps"].get("identical_to_the_oracle_on_the_sample") is True
        if isinstance(line.get("end_to_end_config5"), dict):
            checks["end_to_end_config5_md5s_are_the_references"] = line["end_to_end_config5"].get("product_md5s_are_the_references") is True
        line["self_checks"] = checks
        line["self_checks_all_true"] = all(checks.values())
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        failed = not line["self_checks_all_true"]
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()
    if rank == 0 and failed:
        sys.stderr.write("bench.py: a self-check of the line is not true: %s\n" % [k for k, v in line["self_checks"].items() if not v])
        sys.exit(1)


if __name__ == "__main__":
    main()
