"""BAM + BAI writer for the synthetic reads of synth.py (host-side data tooling).

Needed because the GPU box receives only this repository: the compiled reference
binary used as the checker and our own host driver both read BAM through an index.
Formats: SAM/BAM specification sections 4.1-4.2 (BGZF, records) and 5.2 (BAI).
"""
import struct
import zlib

import numpy as np

_SEQ_CODE = np.zeros(256, dtype=np.uint8)
_SEQ_CODE[:] = 15
for _ch, _v in ((b"A", 1), (b"C", 2), (b"G", 4), (b"T", 8), (b"N", 15)):
    _SEQ_CODE[_ch[0]] = _v
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b

_EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class _Bgzf:
    def __init__(self, fh):
        self.fh = fh
        self.buf = bytearray()
        self.coff = 0

    def tell(self):
        return (self.coff << 16) | len(self.buf)

    def _flush(self):
        if not self.buf:
            return
        data = bytes(self.buf)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        bsize = len(comp) + 25
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
        blk = hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
        self.fh.write(blk)
        self.coff += len(blk)
        self.buf = bytearray()

    def write_atomic(self, data):
        """Append without splitting `data` across blocks (records are small)."""
        if len(self.buf) + len(data) > 0xFF00:
            self._flush()
        self.buf += data
        if len(self.buf) > 0xFF00:
            self._flush()

    def close(self):
        self._flush()
        self.fh.write(_EOF_BLOCK)


def write_bam(path, contigs, rd, qname_prefix="r"):
    """contigs: list of (name, length); rd: synth.Reads (coordinate sorted per contig).
    Writes path and path + '.bai'."""
    n_ref = len(contigs)
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    with open(path, "wb") as fh:
        z = _Bgzf(fh)
        hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", n_ref)
        for name, ln in contigs:
            hdr += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
        z.write_atomic(hdr)
        z._flush()
        # per-reference index state
        bins = [dict() for _ in range(n_ref)]
        lin = [dict() for _ in range(n_ref)]
        L = rd.read_len
        qual = b"\x28" * L
        last = {}
        for i in range(rd.n):
            tid = int(rd.tid[i]); pos = int(rd.pos[i]); flag = int(rd.flag[i])
            unm = bool(flag & 0x4)
            ops = [] if unm else [(int(rd.cig_len[i, j]), int(rd.cig_op[i, j])) for j in range(int(rd.ncig[i]))]
            span = sum(l for l, o in ops if o in (0, 2, 3, 7, 8))
            end = pos + span if span > 0 else pos + 1
            b = reg2bin(pos, end)
            seq = rd.seq[i]
            if unm and not (flag & 0x20):
                seq = _COMP[seq[::-1]]          # stored as sequenced; the caller flips it back (src/indelminer.c:404-409)
            codes = _SEQ_CODE[seq]
            if L & 1:
                codes = np.concatenate([codes, [0]])
            packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()
            qn = ("%s%d" % (qname_prefix, int(rd.pair_id[i]))).encode() + b"\0"
            mapq = 0 if unm else int(rd.mapq)
            mq = 0 if (flag & 0x8) else int(rd.mapq)
            tags = b"MQC" + bytes([mq])
            if getattr(rd, "rg_names", None) is not None:      # optional read groups: rd.rg_names[rd.rg_idx[i]], "" = no RG tag
                name = rd.rg_names[int(rd.rg_idx[i])]
                if name:
                    tags += b"RGZ" + name.encode() + b"\0"
            mtid = tid
            ov = getattr(rd, "overrides", None)                # optional per-record overrides for fuzzing: {i: {field: value}}
            if ov is not None and i in ov:
                o = ov[i]
                tags = o.get("tags", tags); mtid = o.get("mtid", mtid); mapq = o.get("mapq", mapq)
                ops = o.get("ops", ops); packed = o.get("packed", packed)
            cig = b"".join(struct.pack("<I", (l << 4) | o) for l, o in ops)
            body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), mapq, b, len(ops), flag, L, mtid,
                               int(rd.mpos[i]), int(rd.isize[i])) + qn + cig + packed + qual + tags
            rec = struct.pack("<i", len(body)) + body
            if len(z.buf) + len(rec) > 0xFF00:
                z._flush()
            v0 = z.tell()
            z.write_atomic(rec)
            # binning index: one chunk per run of records in the same bin.  A chunk ends where the
            # NEXT record starts (what a reader's tell() shows after the last record of the chunk,
            # also across a block boundary; samtools asserts on it, bam_index.c:695)
            if "chunk" in last:
                last["chunk"][1] = v0
            key = (tid, b)
            if last.get("key") == key:
                pass
            else:
                bins[tid].setdefault(b, []).append([v0, v0])
                last["key"] = key
            last["chunk"] = bins[tid][b][-1]
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                if w not in lin[tid]:
                    lin[tid][w] = v0
        z._flush()
        if "chunk" in last:
            last["chunk"][1] = z.tell()
        z.close()
    with open(path + ".bai", "wb") as fh:
        fh.write(b"BAI\1" + struct.pack("<i", n_ref))
        for t in range(n_ref):
            fh.write(struct.pack("<i", len(bins[t])))
            for b in sorted(bins[t]):
                fh.write(struct.pack("<Ii", b, len(bins[t][b])))
                for u, v in bins[t][b]:
                    fh.write(struct.pack("<QQ", u, v))
            n_intv = (max(lin[t]) + 1) if lin[t] else 0
            fh.write(struct.pack("<i", n_intv))
            prev = 0
            for w in range(n_intv):
                prev = lin[t].get(w, prev)       # fill_missing: empty windows inherit the previous offset
                fh.write(struct.pack("<Q", prev))


def write_fasta(path, contigs, refs, width=60):
    with open(path, "wb") as fh:
        for (name, _), ref in zip(contigs, refs):
            fh.write((">%s\n" % name).encode())
            ref = np.asarray(ref, dtype=np.uint8)
            full = len(ref) // width * width
            if full:                        # whole lines in one go: rows of `width` bases + a newline column
                rows = np.empty((full // width, width + 1), dtype=np.uint8)
                rows[:, :width] = ref[:full].reshape(-1, width)
                rows[:, width] = 10
                fh.write(rows.tobytes())
            if full < len(ref):
                fh.write(ref[full:].tobytes() + b"\n")
