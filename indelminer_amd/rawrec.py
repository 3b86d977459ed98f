"""BAM record bytes of synthetic reads, built with numpy (host-side data tooling, no compute path).

Two consumers:
  * records(rd, ...)      the DEVICE layout im_dev_triage reads (include/indelminer_amd.h: the 32-byte
                          core + variable part of every record, no block_size word, 4-byte aligned starts)
                          -- what the host driver ships to the GPU, made here directly for bench.py / tests;
  * write_bam_fast(...)   the same records as a coordinate-sorted BAM + BAI, for whole-program runs at
                          sizes where bamwrite.write_bam's per-record Python loop takes minutes.
Record layout: SAM/BAM specification 4.2; identical field values to bamwrite.write_bam.
"""
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import bamwrite

_SEQ_CODE = bamwrite._SEQ_CODE
_COMP = bamwrite._COMP

CORE_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("l_qname", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                       ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("mtid", "<i4"), ("mpos", "<i4"),
                       ("isize", "<i4")])
assert CORE_DTYPE.itemsize == 32


def reg2bin_vec(beg, end):
    end = end - 1
    out = np.zeros(len(beg), dtype=np.int64)
    done = np.zeros(len(beg), dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        m = ~done & ((beg >> shift) == (end >> shift))
        out[m] = base + (beg[m] >> shift)
        done |= m
    return out


def _scatter(dst, starts, lens, flat):
    """dst[starts[i] : starts[i] + lens[i]] = consecutive slices of flat."""
    tot = int(lens.sum())
    if tot == 0:
        return
    excl = np.cumsum(lens) - lens
    idx = np.repeat(starts - excl, lens) + np.arange(tot, dtype=np.int64)
    dst[idx] = flat


def build(rd, lo=0, hi=None, qname_prefix="r", align=4, block_size_word=False, rg=None, aux_prefix=b""):
    """Records lo..hi of rd as one uint8 array.  Returns (raw, off) with off[n + 1] (int64) the record
    starts (of the block_size word when block_size_word).  rg: optional read-group name (RG:Z tag); aux_prefix: bytes of whole aux
    fields put in front of the MQ tag of every record (what an aligner writes first: NM, MD, AS, XS ...)."""
    hi = rd.n if hi is None else hi
    n = hi - lo
    L = rd.read_len
    sl = slice(lo, hi)
    flag = rd.flag[sl].astype(np.int64)
    unm = (flag & 0x4) != 0
    ncig = np.where(unm, 0, rd.ncig[sl].astype(np.int64))
    pos = rd.pos[sl].astype(np.int64)
    valid = np.arange(rd.cig_op.shape[1])[None, :] < ncig[:, None]
    refop = np.isin(rd.cig_op[sl], (0, 2, 3, 7, 8)) & valid
    span = (rd.cig_len[sl].astype(np.int64) * refop).sum(1)
    end = np.where(span > 0, pos + span, pos + 1)
    names = np.char.add(qname_prefix, rd.pair_id[sl].astype(np.int64).astype(str)).astype("S")
    qlen = np.char.str_len(names).astype(np.int64) + 1
    tags_fixed = 4
    rg_bytes = b"" if rg is None else b"RGZ" + rg.encode() + b"\0"
    body = 32 + qlen + 4 * ncig + (L + 1) // 2 + L + len(aux_prefix) + tags_fixed + len(rg_bytes)
    pre = 4 if block_size_word else 0
    rec = pre + body
    step = (rec + align - 1) // align * align
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(step, out=off[1:])
    raw = np.zeros(int(off[-1]), dtype=np.uint8)
    start = off[:-1] + pre

    core = np.zeros(n, dtype=CORE_DTYPE)
    core["tid"] = rd.tid[sl]; core["pos"] = pos; core["l_qname"] = qlen
    core["mapq"] = np.where(unm, 0, rd.mapq); core["bin"] = reg2bin_vec(pos, end)
    core["n_cigar"] = ncig; core["flag"] = flag; core["l_seq"] = L
    core["mtid"] = rd.tid[sl]; core["mpos"] = rd.mpos[sl]; core["isize"] = rd.isize[sl]
    raw[(start[:, None] + np.arange(32)[None, :]).reshape(-1)] = core.view(np.uint8).reshape(-1)
    if block_size_word:
        raw[(off[:-1, None] + np.arange(4)[None, :]).reshape(-1)] = body.astype("<i4").view(np.uint8).reshape(-1)

    # qname (NUL terminated: the buffer is zero-filled)
    width = names.dtype.itemsize
    nm = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, width)
    mask = np.arange(width)[None, :] < (qlen - 1)[:, None]
    _scatter(raw, start + 32, qlen - 1, nm[mask])
    # cigar
    cw = ((rd.cig_len[sl].astype(np.int64) << 4) | rd.cig_op[sl].astype(np.int64)).astype("<u4")
    cbytes = cw.view(np.uint8).reshape(n, -1, 4)[valid].reshape(-1)
    o_cig = start + 32 + qlen
    _scatter(raw, o_cig, 4 * ncig, cbytes)
    # seq: stored as sequenced for an unmapped read whose mate is forward (bamwrite.write_bam)
    seq = rd.seq[sl]
    flip = unm & ((flag & 0x20) == 0)
    if flip.any():
        seq = seq.copy()
        seq[flip] = _COMP[seq[flip][:, ::-1]]
    codes = _SEQ_CODE[seq]
    if L & 1:
        codes = np.concatenate([codes, np.zeros((n, 1), np.uint8)], axis=1)
    packed = ((codes[:, 0::2] << 4) | codes[:, 1::2]).astype(np.uint8)
    o_seq = o_cig + 4 * ncig
    hb = (L + 1) // 2
    raw[(o_seq[:, None] + np.arange(hb)[None, :]).reshape(-1)] = packed.reshape(-1)
    o_qual = o_seq + hb
    raw[(o_qual[:, None] + np.arange(L)[None, :]).reshape(-1)] = 0x28
    o_tag = o_qual + L
    if aux_prefix:
        ab = np.frombuffer(bytes(aux_prefix), dtype=np.uint8)
        raw[(o_tag[:, None] + np.arange(len(ab))[None, :]).reshape(-1)] = np.tile(ab, n)
        o_tag = o_tag + len(ab)
    mq = np.where((flag & 0x8) != 0, 0, rd.mapq).astype(np.uint8)
    tag = np.zeros((n, 4), dtype=np.uint8)
    tag[:, 0] = ord("M"); tag[:, 1] = ord("Q"); tag[:, 2] = ord("C"); tag[:, 3] = mq
    raw[(o_tag[:, None] + np.arange(4)[None, :]).reshape(-1)] = tag.reshape(-1)
    if rg_bytes:
        rb = np.frombuffer(rg_bytes, dtype=np.uint8)
        raw[((o_tag + 4)[:, None] + np.arange(len(rb))[None, :]).reshape(-1)] = np.tile(rb, n)
    return raw, off, dict(pos=pos, end=end, bin=core["bin"].astype(np.int64), tid=rd.tid[sl].astype(np.int64))


def records(rd, lo=0, hi=None, qname_prefix="r", rg=None, qual=True, aux_prefix=b""):
    """Device layout: (raw uint8, rec_off uint32[n + 1]).  qual=False: without base qualities, as the product delivers records."""
    raw, off, _ = build(rd, lo, hi, qname_prefix, align=4, block_size_word=False, rg=rg, aux_prefix=aux_prefix)
    assert off[-1] < 2**32
    if not qual:
        return strip_quals(raw, off)
    return raw, off.astype(np.uint32)


def strip_quals(raw, off):
    """The device layout WITHOUT base qualities, as the product's walkers deliver records (include/indelminer_amd.h, im_dev_records):
    (qname, cigar, seq, aux) behind the core, bin = 0xFFFF -- except records whose CIGAR asks for more read bases than l_seq, which
    stay as they are.  Returns (raw, rec_off uint32[n + 1])."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    off = np.asarray(off, dtype=np.int64)
    n = len(off) - 1
    if n == 0:
        return raw.copy(), off.astype(np.uint32)
    start, end = off[:-1], off[1:]
    ok = end - start >= 32
    base = np.where(ok, start, 0)
    u8 = lambda o: raw[base + o].astype(np.int64)
    l_qname = u8(8)
    n_cigar = u8(12) | (u8(13) << 8)
    l_seq = (u8(16) | (u8(17) << 8) | (u8(18) << 16) | (u8(19) << 24)).astype(np.int64)
    l_seq = np.where(l_seq >= 1 << 31, l_seq - (1 << 32), l_seq)
    head = 32 + l_qname + 4 * n_cigar
    packed = (np.maximum(l_seq, 0) + 1) >> 1
    fits = ok & (l_seq >= 0) & (head + packed + l_seq <= end - start)
    qlen = np.zeros(n, np.int64)
    for k in range(int(n_cigar[fits].max()) if fits.any() else 0):
        has = fits & (n_cigar > k)
        o = np.where(has, base + 32 + l_qname + 4 * k, 0)
        cw = raw[o].astype(np.int64) | (raw[o + 1].astype(np.int64) << 8) | (raw[o + 2].astype(np.int64) << 16) | (raw[o + 3].astype(np.int64) << 24)
        op = cw & 15
        qlen += np.where(has & np.isin(op, (0, 1, 4, 7, 8)), cw >> 4, 0)
    drop = fits & (qlen <= l_seq)
    cut = np.where(drop, l_seq, 0)
    # real length of each record without its alignment padding is unknown here: keep the bytes behind the qualities as they are
    new_len = (end - start) - cut
    new_step = (new_len + 3) // 4 * 4
    new_off = np.zeros(n + 1, np.int64)
    np.cumsum(new_step, out=new_off[1:])
    out = np.zeros(int(new_off[-1]) + 64, np.uint8)
    first = np.where(drop, head + packed, end - start)          # bytes in front of the qualities
    _scatter(out, new_off[:-1], first, raw[np.repeat(start - (np.cumsum(first) - first), first) + np.arange(int(first.sum()), dtype=np.int64)])
    rest = (end - start) - first - cut
    src0 = start + first + cut
    _scatter(out, new_off[:-1] + first, rest, raw[np.repeat(src0 - (np.cumsum(rest) - rest), rest) + np.arange(int(rest.sum()), dtype=np.int64)])
    dropped = np.nonzero(drop)[0]
    out[new_off[dropped] + 10] = 0xFF; out[new_off[dropped] + 11] = 0xFF
    kept_marker = np.nonzero(~drop & ok & (raw[base + 10] == 0xFF) & (raw[base + 11] == 0xFF))[0]
    out[new_off[kept_marker] + 10] = 0xFE
    return out[:int(new_off[-1])].copy(), new_off.astype(np.uint32)


def records_from_bam(path):
    """Device layout of every record of a BAM file, in file order (small files: tests and goldens).
    Returns (raw uint8, rec_off uint32[n + 1], contigs [(name, length)])."""
    data = open(path, "rb").read()
    out = bytearray()
    p = 0
    while p < len(data):                          # BGZF = concatenated gzip members
        d = zlib.decompressobj(31)
        out += d.decompress(data[p:])
        used = len(data) - p - len(d.unused_data)
        if used <= 0:
            break
        p += used
    buf = bytes(out)
    assert buf[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", buf, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]; p += 4
    contigs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", buf, p)[0]; p += 4
        name = buf[p:p + l_name - 1].decode(); p += l_name
        contigs.append((name, struct.unpack_from("<i", buf, p)[0])); p += 4
    pieces, off = [], [0]
    while p + 4 <= len(buf):
        bs = struct.unpack_from("<i", buf, p)[0]; p += 4
        body = buf[p:p + bs]; p += bs
        pad = (-len(body)) % 4
        pieces.append(body + b"\0" * pad)
        off.append(off[-1] + len(body) + pad)
    raw = np.frombuffer(b"".join(pieces), dtype=np.uint8).copy() if pieces else np.zeros(0, np.uint8)
    return raw, np.asarray(off, dtype=np.uint32), contigs


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def write_bam_fast(path, contigs, rd, qname_prefix="r", level=1, chunk=1 << 20, threads=8):
    """Same file content model as bamwrite.write_bam (records never split across BGZF blocks), built from
    vectorised record bytes; blocks are deflated on a thread pool.  Writes path and path + '.bai'."""
    n_ref = len(contigs)
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", n_ref)
    for name, ln in contigs:
        hdr += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    voff = np.zeros(rd.n + 1, dtype=np.uint64)          # virtual offset of every record (+ end of data)
    meta = dict(pos=[], end=[], bin=[], tid=[])
    with open(path, "wb") as fh, ThreadPoolExecutor(threads) as pool:
        blk = _bgzf_block(hdr, level)
        fh.write(blk)
        coff = len(blk)
        for lo in range(0, rd.n, chunk):
            hi = min(rd.n, lo + chunk)
            raw, off, m = build(rd, lo, hi, qname_prefix, align=1, block_size_word=True)
            for k in meta:
                meta[k].append(m[k])
            # greedy blocks of whole records, <= 0xFF00 payload bytes
            cuts = [0]
            while cuts[-1] < hi - lo:
                s = cuts[-1]
                e = int(np.searchsorted(off, off[s] + 0xFF00, side="right")) - 1
                cuts.append(max(e, s + 1))
            pieces = [raw[off[cuts[i]]:off[cuts[i + 1]]].tobytes() for i in range(len(cuts) - 1)]
            comp = list(pool.map(lambda d: _bgzf_block(d, level), pieces))
            for i, c in enumerate(comp):
                s, e = cuts[i], cuts[i + 1]
                voff[lo + s:lo + e] = (np.uint64(coff) << np.uint64(16)) | (off[s:e] - off[s]).astype(np.uint64)
                fh.write(c)
                coff += len(c)
        voff[rd.n] = np.uint64(coff) << np.uint64(16)
        fh.write(bamwrite._EOF_BLOCK)
    pos = np.concatenate(meta["pos"]) if rd.n else np.zeros(0, np.int64)
    end = np.concatenate(meta["end"]) if rd.n else np.zeros(0, np.int64)
    bins = np.concatenate(meta["bin"]) if rd.n else np.zeros(0, np.int64)
    tids = np.concatenate(meta["tid"]) if rd.n else np.zeros(0, np.int64)
    with open(path + ".bai", "wb") as fh:
        fh.write(b"BAI\1" + struct.pack("<i", n_ref))
        # runs of records in the same (tid, bin): one chunk each, ending where the next record starts
        if rd.n:
            change = np.concatenate([[True], (tids[1:] != tids[:-1]) | (bins[1:] != bins[:-1])])
            run_start = np.nonzero(change)[0]
            run_end = np.concatenate([run_start[1:], [rd.n]])
        else:
            run_start = run_end = np.zeros(0, np.int64)
        for t in range(n_ref):
            sel = np.nonzero(tids[run_start] == t)[0] if rd.n else np.zeros(0, np.int64)
            by_bin = {}
            for r in sel:
                by_bin.setdefault(int(bins[run_start[r]]), []).append((int(voff[run_start[r]]), int(voff[run_end[r]])))
            fh.write(struct.pack("<i", len(by_bin)))
            for b in sorted(by_bin):
                fh.write(struct.pack("<Ii", b, len(by_bin[b])))
                for u, v in by_bin[b]:
                    fh.write(struct.pack("<QQ", u, v))
            m = tids == t
            if m.any():
                w0 = pos[m] >> 14
                w1 = (end[m] - 1) >> 14
                n_intv = int(w1.max()) + 1
                lin = np.full(n_intv, np.iinfo(np.uint64).max, dtype=np.uint64)
                vo = voff[:-1][m]
                # records are coordinate sorted: the first record touching a window has the smallest offset
                for w in (w0, w1):
                    np.minimum.at(lin, w, vo)
                span = w1 - w0
                for r in np.nonzero(span > 1)[0]:                      # reads longer than a window: none at 100 bp
                    lin[w0[r] + 1:w1[r]] = np.minimum(lin[w0[r] + 1:w1[r]], vo[r])
                prev = np.uint64(0)
                fh.write(struct.pack("<i", n_intv))
                out = np.zeros(n_intv, dtype=np.uint64)
                for w in range(n_intv):
                    if lin[w] != np.iinfo(np.uint64).max:
                        prev = lin[w]
                    out[w] = prev
                fh.write(out.astype("<u8").tobytes())
            else:
                fh.write(struct.pack("<i", 0))
