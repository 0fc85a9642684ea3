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
    # ---- the other SV types from the SAME real split reads (round 4): the example data only carries a deletion, so the
    # reference is rearranged instead of the reads.  A = the 600 real bases left of the deletion (chr18 99 400 - 100 000),
    # C = the 600 right of it (108 001 - 108 601); a split read of the example joins the end of A to the start of C.  Placed
    # in small chromosomes made of other real chr18 sequence (X, Y, Z), the same reads support
    #   DUP (svt 3):       chr' = X C Y A Z          the junction joins the END of the duplicated span to its START
    #   INV 3to3 (svt 0):  chr' = X A Y rc(C) Z      left segment forward, right segment reverse-complemented
    #   INV 5to5 (svt 1):  chr' = X rc(A) Y C Z      left segment reverse-complemented, right segment forward
    #   BND 3to5 / 5to3 / 3to3 / 5to5 (svt 7, 8, 5, 6): two chromosomes, the second part built into svRefStr (src/split.h:73-113)
    # plus, per type, candidates whose "split reads" are ordinary reads (the reference answers false or true; compared as is).
    # Reads enter the set in the junction's orientation, i.e. as src/shortpe.h:124-137 + _adjustOrientation (src/split.h:55-68)
    # leave them; one candidate per inversion type gets them reverse-complemented (what a caller that skipped that step would
    # hand over: the reference answers false).
    from delly_amd import synth
    chr18 = synth.load_real_chromosome().tobytes()
    comp = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")

    def rc(x):
        return x.translate(comp)[::-1]

    A, C = chr18[99400:100000], chr18[108001:108601]
    X, Y, Z = chr18[20000:22000], chr18[40000:43000], chr18[60000:62000]
    _, recs = bamlite.read_bam(os.path.join(EX, "sr.bam"))
    recs = [r for r in recs if not (r.flag & 0x904) and r.ref_id == 0]
    split_reads = list(dict.fromkeys(r.seq for r in recs if (r.right_clip() >= 10 and abs(r.ref_end() - 100000) <= 2) or
                                     (r.left_clip() >= 10 and abs(r.pos - 108001) <= 2)))[:20]
    plain = list(dict.fromkeys(r.seq for r in recs if 20200 <= r.pos <= 20700 and not r.left_clip() and not r.right_clip()))[:12]
    rcs = lambda reads: [rc(x.encode()).decode() for x in reads]
    chroms, groups = [], []

    def add(svt, seqs, chr_, chr2, start, end, reads):
        base = len(chroms)
        chroms.extend(np.frombuffer(x, dtype=np.uint8) for x in seqs)
        groups.append((svt, base + chr_, base + chr2, start, end, reads))

    lx, ly = len(X), len(Y)
    add(3, [X + C + Y + A + Z], 0, 0, lx, lx + 600 + ly + 600, split_reads)                         # DUP: start = first base of C, end = behind A
    add(0, [X + A + Y + rc(C) + Z], 0, 0, lx + 600, lx + 600 + ly + 600, split_reads)                 # INV 3to3
    add(1, [X + rc(A) + Y + C + Z], 0, 0, lx, lx + 600 + ly, split_reads)                             # INV 5to5: rc(chr'[s ..)) then chr'[e ..)
    add(7, [X + A + Z, Y + C + Z], 0, 1, lx + 600, ly, split_reads)                                   # BND 3to5
    add(8, [X + C + Z, Y + A + Z], 0, 1, lx, ly + 600, split_reads)                                   # BND 5to3
    add(5, [X + A + Z, Y + rc(C) + Z], 0, 1, lx + 600, ly + 600, split_reads)                         # BND 3to3
    add(6, [X + rc(A) + Z, Y + C + Z], 0, 1, lx, ly, split_reads)                                     # BND 5to5
    add(0, [X + A + Y + rc(C) + Z], 0, 0, lx + 600, lx + 600 + ly + 600, rcs(split_reads))            # the wrong strand
    add(1, [X + rc(A) + Y + C + Z], 0, 0, lx, lx + 600 + ly, rcs(split_reads))
    for svt in (0, 1, 3, 5, 6, 7, 8):                                                                 # ordinary reads as candidates of every type
        two = svt >= 5
        add(svt, [X + A + Z, Y + C + Z] if two else [X + A + Y + C + Z], 0, 1 if two else 0, 700, 900 if two else 700 + 2600, plain)
    junc = np.zeros(len(groups), dtype=abi.junction_dtype())
    seqs = []
    for k, (svt, c1, c2, start, end, reads) in enumerate(groups):
        junc[k]["svid"] = k
        junc[k]["svt"] = svt
        junc[k]["chr"] = c1
        junc[k]["chr2"] = c2
        junc[k]["sv_start"] = start
        junc[k]["sv_end"] = end
        junc[k]["seq_first"] = len(seqs)
        junc[k]["n_seq"] = len(reads)
        seqs.extend(np.frombuffer(r.encode(), dtype=np.uint8) for r in reads)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs], dtype=np.uint64)
    out["svx_junc"], out["svx_blob"], out["svx_off"] = junc, np.concatenate(seqs), off
    out["svx_nchr"] = np.int64(len(chroms))
    for i, ch in enumerate(chroms):
        out["svx_chr%d" % i] = ch
    print("svx: %d junctions of svt %s on %d rearranged chromosomes, %d split reads" % (len(groups), sorted(set(g[0] for g in groups)), len(chroms), len(split_reads)))
    np.savez_compressed(os.path.join(HERE, "example_reads.npz"), **out)
    print("example_reads.npz: %d bytes" % os.path.getsize(os.path.join(HERE, "example_reads.npz")))


if __name__ == "__main__":
    main()
