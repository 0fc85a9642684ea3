"""Regenerates tests/golden/example_reads.npz from the reference's example BAMs (BASELINE configs[0] fixtures; the GPU box has
neither /root/reference nor htslib).  Reads are REAL: 150 bp Illumina-like reads of example/sr.bam and ONT reads of
example/lr.bam, both against example/ref.fa (= tests/golden/chr18_example.npz), which carries one 8 kb deletion
(100 000 - 108 001).  Three junction batches in the layout of include/dellyhip.h:

  sr     : the deletion with its soft-clipped split reads (what src/shortpe.h:96-156 collects), plus candidate junctions at
           regular positions whose "split reads" are the ordinary reads covering that position (false candidates: real base
           errors for msa(), alignConsensus false or true as the reference decides).
  lr     : the deletion with the slices (+- window around the breakpoint, src/assemble.h:807-831) of the long reads that
           support it -- one read with an 8 kb D operation, the others soft-clipped at either breakpoint -- plus candidates at
           regular positions with >= 2 kb slices of the reads covering them (msaEdlib + alignConsensus(realign)).
  lrins  : the same slices as insertion candidates (svt 4): msaWfa + splitAlign.

Run in the dev container:  python tests/golden/make_example_reads.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bamlite  # noqa: E402
from delly_amd import abi  # noqa: E402

EX = "/root/reference/example"


def read_offset(rec, ref_pos):
    """index into rec.seq of the base aligned at (or first after) reference position ref_pos; None if not covered"""
    q, p = 0, rec.pos
    for op, n in rec.cigar:
        if op in "M=X":
            if p <= ref_pos < p + n:
                return q + (ref_pos - p)
            q += n
            p += n
        elif op in "IS":
            q += n
        elif op in "DN":
            if p <= ref_pos < p + n:
                return q
            p += n
    return None


def pack(groups, svt, with_positions):
    junc = np.zeros(len(groups), dtype=abi.junction_dtype())
    seqs = []
    for k, (start, end, reads, ins_len) in enumerate(groups):
        junc[k]["svid"] = k
        junc[k]["svt"] = svt
        junc[k]["sv_start"] = start
        junc[k]["sv_end"] = end
        junc[k]["ins_len"] = ins_len
        junc[k]["seq_first"] = len(seqs)
        junc[k]["n_seq"] = len(reads)
        seqs.extend(np.frombuffer(r.encode(), dtype=np.uint8) for r in reads)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([s.size for s in seqs], dtype=np.uint64)
    return junc, (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)), off


def main():
    out = {}
    # ---- short reads
    _, recs = bamlite.read_bam(os.path.join(EX, "sr.bam"))
    recs = [r for r in recs if not (r.flag & 0x904) and r.ref_id == 0]
    del_reads = [r.seq for r in recs if (r.right_clip() >= 10 and abs(r.ref_end() - 100000) <= 2) or
                 (r.left_clip() >= 10 and abs(r.pos - 108001) <= 2)]
    assert len(del_reads) >= 8, len(del_reads)
    groups = [(99995, 108004, list(dict.fromkeys(del_reads))[:20], 0)]
    for p in range(7000, 195000, 9000):
        if 95000 < p < 112000:
            continue
        cov = [r.seq for r in recs if r.pos + 20 <= p <= r.ref_end() - 20 and not r.left_clip() and not r.right_clip()]
        cov = list(dict.fromkeys(cov))[:20]       # (a std::unordered_set collapses duplicates, src/shortpe.h:68)
        if len(cov) >= 3:
            groups.append((p, p + 600, cov, 0))
    out["sr_junc"], out["sr_blob"], out["sr_off"] = pack(groups, 2, True)
    print("sr: %d junctions, %d reads (the deletion: %d split reads)" % (len(groups), sum(len(g[2]) for g in groups), len(groups[0][2])))
    # ---- long reads
    _, recs = bamlite.read_bam(os.path.join(EX, "lr.bam"))
    recs = [r for r in recs if not (r.flag & 0x904) and r.ref_id == 0]
    W = 1100

    def slices(p, need_clip=None):
        got = []
        for r in recs:
            q = None
            if need_clip:
                if r.right_clip() >= 200 and abs(r.ref_end() - 100000) <= 60:
                    q = len(r.seq) - r.right_clip()
                elif r.left_clip() >= 200 and abs(r.pos - 108000) <= 60:
                    q = r.left_clip()
                elif r.pos < 99000 and r.ref_end() > 109000:
                    q = read_offset(r, 100000)
            elif r.pos + 300 <= p <= r.ref_end() - 300 and r.left_clip() < 50 and r.right_clip() < 50:
                q = read_offset(r, p)
            if q is None:
                continue
            s = r.seq[max(0, q - W):q + W]
            if len(s) >= 2000:
                got.append(s)
        return got[:15]

    lr_groups = [(100000, 108000, slices(100000, need_clip=True), 0)]
    assert len(lr_groups[0][2]) >= 3, len(lr_groups[0][2])
    for p in range(12000, 190000, 16000):
        if 90000 < p < 118000:
            continue
        s = slices(p)
        if len(s) >= 3:
            lr_groups.append((p, p + 3000, s, 0))
    out["lr_junc"], out["lr_blob"], out["lr_off"] = pack(lr_groups, 2, True)
    print("lr: %d junctions, %d read slices (the deletion: %d)" % (len(lr_groups), sum(len(g[2]) for g in lr_groups), len(lr_groups[0][2])))
    ins_groups = [(g[0], g[0] + 1, g[2], 300) for g in lr_groups[1:]]
    out["lrins_junc"], out["lrins_blob"], out["lrins_off"] = pack(ins_groups, 4, True)
    np.savez_compressed(os.path.join(HERE, "example_reads.npz"), **out)
    print("example_reads.npz: %d bytes" % os.path.getsize(os.path.join(HERE, "example_reads.npz")))


if __name__ == "__main__":
    main()
