"""CPU checks of the triage oracle (oracle/im_oracle_triage.c) that need no device: the record
contract of include/indelminer_amd.h (im_dev_records)."""
import struct

import numpy as np

from tests.support import oraclebind as ob


def _rec(flag, cigar, tags=b"", qname=b"q\0", pad=b"\0\0\0", l_seq=100):
    cig = b"".join(struct.pack("<I", (l << 4) | o) for l, o in cigar)
    body = struct.pack("<iiBBHHHiiii", 0, 100, len(qname), 60, 0, len(cigar), flag, l_seq, 0, 300, 300) + qname + cig \
        + bytes([0x12] * ((l_seq + 1) // 2)) + b"\x28" * l_seq + tags
    return body + pad[:(-len(body)) % 4]


def test_alignment_padding_is_not_an_aux_field():
    """records sit at 4-byte aligned offsets, so up to three spare bytes follow the aux area; whatever they hold
    (here the start of an RG / MQ field) must not be read as a tag -- found on a 24-contig BAM where the pinned
    staging buffer's previous contents spelled 'RG' behind a record"""
    S = ((30, 4), (70, 0))
    recs = [_rec(0x3, S, qname=b"qq\0", tags=b"MQC\x0a", pad=b"RGZ"), _rec(0x3, S, qname=b"qq\0", tags=b"MQC\x0a"),
            _rec(0x3, S, qname=b"qqq\0", pad=b"MQ"), _rec(0x3, S, qname=b"qqq\0")]
    assert [len(r) % 4 for r in recs] == [0, 0, 0, 0]
    raw = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    off = np.zeros(len(recs) + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in recs], out=off[1:])
    tri = ob.triage_records(raw, off, ["generic"], [700])
    assert tri[0][0].cls == tri[1][0].cls == 3 and tri[2][0].cls == tri[3][0].cls == 3
    assert tri[0][0].range_max == 700
