"""Minimal BAM reader used only by the tests (fixture generation / parity inputs).

BGZF is a series of gzip members, so zlib can inflate it member by member.
Record layout: SAM/BAM specification section 4.2.
"""
import struct
import zlib

SEQ_DECODE = "=ACMGRSVTWYHKDBN"
CIGAR_CHARS = "MIDNSHP=XB"


def bgzf_decompress(path):
    data = open(path, "rb").read()
    out = bytearray()
    pos = 0
    while pos < len(data):
        d = zlib.decompressobj(31)
        out += d.decompress(data[pos:])
        used = len(data) - pos - len(d.unused_data)
        if used <= 0:
            break
        pos += used
    return bytes(out)


class Record:
    __slots__ = ("tid", "pos", "mapq", "flag", "l_seq", "mtid", "mpos", "isize",
                 "qname", "cigar", "seq", "tags")

    def aux_int(self, tag):
        v = self.tags.get(tag)
        return v[1] if v is not None and v[0] in "cCsSiI" else None

    def aux_str(self, tag):
        v = self.tags.get(tag)
        return v[1] if v is not None and v[0] == "Z" else None


def _parse_tags(buf):
    tags = {}
    i = 0
    n = len(buf)
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while i + 3 <= n:
        tag = buf[i:i + 2].decode()
        t = chr(buf[i + 2])
        i += 3
        if t in fmt:
            sz = struct.calcsize(fmt[t])
            tags[tag] = (t, struct.unpack_from(fmt[t], buf, i)[0])
            i += sz
        elif t == "A":
            tags[tag] = (t, chr(buf[i])); i += 1
        elif t in "ZH":
            j = buf.index(b"\0", i)
            tags[tag] = (t, buf[i:j].decode()); i = j + 1
        elif t == "B":
            st = chr(buf[i]); cnt = struct.unpack_from("<I", buf, i + 1)[0]
            sz = struct.calcsize(fmt[st])
            tags[tag] = (t, None); i += 5 + sz * cnt
        else:
            raise ValueError("bad aux type %r" % t)
    return tags


def read_bam(path):
    """Returns (header_text, [(name, length)], [Record...]) in file order."""
    raw = bgzf_decompress(path)
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode(errors="replace")
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]; p += 4
        name = raw[p:p + l_name - 1].decode(); p += l_name
        l_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
        refs.append((name, l_ref))
    recs = []
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]; p += 4
        (tid, pos, l_qname, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, isize) = \
            struct.unpack_from("<iiBBHHHiiii", raw, p)
        q = p + 32
        r = Record()
        r.tid, r.pos, r.mapq, r.flag, r.l_seq = tid, pos, mapq, flag, l_seq
        r.mtid, r.mpos, r.isize = mtid, mpos, isize
        r.qname = raw[q:q + l_qname - 1].decode(); q += l_qname
        r.cigar = [(c >> 4, c & 15) for c in struct.unpack_from("<%dI" % n_cig, raw, q)]; q += 4 * n_cig
        sb = raw[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = []
        for i in range(l_seq):
            b = sb[i >> 1]
            seq.append(SEQ_DECODE[(b >> 4) if (i & 1) == 0 else (b & 15)])
        r.seq = "".join(seq)
        q += l_seq  # qualities
        r.tags = _parse_tags(raw[q:p + bs])
        recs.append(r)
        p += bs
    return text, refs, recs


def read_fasta(path):
    """Contigs as upper-cased strings, the way read_reference (src/shared.c:46-82) keeps them."""
    names, seqs, cur = [], [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            names.append(line[1:].split()[0]); cur = []; seqs.append(cur)
        elif cur is not None:
            cur.append(line.strip())
    return names, ["".join(s).upper() for s in seqs]
