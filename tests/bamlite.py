"""TEST INFRASTRUCTURE ONLY: a minimal BGZF / BAM record reader (Python gzip + struct, no htslib).  It exists so that
tests/golden/make_example_reads.py can pull real reads out of the reference's example BAMs (example/sr.bam, example/lr.bam:
BASELINE configs[0]) and commit them as parity fixtures.  Not part of the product and not a step towards junction
extraction (SURVEY.md 8f N4 stays out of scope): no index, no CRAM, no tags beyond SA."""
import gzip
import struct

_SEQ = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


class Record:
    __slots__ = ("ref_id", "pos", "mapq", "flag", "cigar", "seq", "name", "sa")

    def ref_end(self):
        return self.pos + sum(n for op, n in self.cigar if op in "MDN=X")

    def left_clip(self):
        return self.cigar[0][1] if self.cigar and self.cigar[0][0] == "S" else 0

    def right_clip(self):
        return self.cigar[-1][1] if self.cigar and self.cigar[-1][0] == "S" else 0


def read_bam(path):
    """-> (list of (name, length) references, generator of Record)"""
    f = gzip.open(path, "rb")      # BGZF is a series of gzip members: the gzip module reads them back to back
    assert f.read(4) == b"BAM\x01"
    (l_text,) = struct.unpack("<i", f.read(4))
    f.read(l_text)
    (n_ref,) = struct.unpack("<i", f.read(4))
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack("<i", f.read(4))
        name = f.read(l_name)[:-1].decode()
        (l_ref,) = struct.unpack("<i", f.read(4))
        refs.append((name, l_ref))

    def records():
        while True:
            head = f.read(4)
            if len(head) < 4:
                return
            (block,) = struct.unpack("<i", head)
            buf = f.read(block)
            ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, _nref, _npos, _tlen = struct.unpack("<iiBBHHHiiii", buf[:32])
            o = 32
            r = Record()
            r.ref_id, r.pos, r.mapq, r.flag = ref_id, pos, mapq, flag
            r.name = buf[o:o + l_read_name - 1].decode()
            o += l_read_name
            cig = struct.unpack("<%dI" % n_cigar, buf[o:o + 4 * n_cigar])
            r.cigar = [(CIGAR_OPS[c & 15], c >> 4) for c in cig]
            o += 4 * n_cigar
            packed = buf[o:o + (l_seq + 1) // 2]
            o += (l_seq + 1) // 2
            seq = []
            for b in packed:
                seq.append(_SEQ[b >> 4])
                seq.append(_SEQ[b & 15])
            r.seq = "".join(seq[:l_seq])
            o += l_seq   # qualities
            r.sa = None
            # optional fields: only SA:Z is looked for
            while o + 3 <= len(buf):
                tag, typ = buf[o:o + 2], chr(buf[o + 2])
                o += 3
                if typ == "Z":
                    e = buf.index(b"\x00", o)
                    if tag == b"SA":
                        r.sa = buf[o:e].decode()
                    o = e + 1
                elif typ in "AcC":
                    o += 1
                elif typ in "sS":
                    o += 2
                elif typ in "iIf":
                    o += 4
                elif typ == "H":
                    o = buf.index(b"\x00", o) + 1
                elif typ == "B":
                    sub = chr(buf[o])
                    (cnt,) = struct.unpack("<i", buf[o + 1:o + 5])
                    o += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
                else:
                    break
            yield r

    return refs, records()
