"""Synthetic junction batches shared by the parity tests, the
oracle and bench.py.  Counter-based: junction j of a batch depends only on
(seed, j), never on the batch size or on the rank that generates it, so
N-GPU shards of one logical batch are reproducible.

A batch mimics what the loop of src/shortpe.h:96-201 holds per chromosome:
one chromosome string (here: the per-junction 4 kb windows concatenated), the
SV candidates with approximate coordinates, and either their consensus
(unit U: alignConsensus only) or their split reads in host iteration order
(unit U_full: msa + alignConsensus).
"""
import numpy as np

from . import abi

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b

WINDOW = 4000  # bases of private genome per junction
WINDOW_LR = 16000  # ... for mode "lr" (BASELINE config C4 shapes)


def revcomp(a):
    return _COMP[a[::-1]]


def _rng(seed, j):
    return np.random.Generator(np.random.Philox(key=[seed, 0x5eed], counter=[j, 0, 0, 0]))


def _ont(rng, seq, rate):
    """ONT-like errors: one third each substitution / insertion / deletion"""
    u = rng.random(seq.size)
    keep = u >= rate / 3                      # deletions
    seq = seq[keep]
    u = u[keep]
    sub = (u >= rate / 3) & (u < 2 * rate / 3)
    seq = seq.copy()
    seq[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    at = np.nonzero(u >= 1 - rate / 3)[0]     # insertions
    return np.insert(seq, at, ACGT[rng.integers(0, 4, at.size)])


def _mutate(rng, seq, rate):
    seq = seq.copy()
    if rate > 0:
        hit = np.nonzero(rng.random(seq.size) < rate)[0]
        for i in hit:
            # substitute with a different base
            cur = seq[i]
            cand = ACGT[ACGT != cur]
            seq[i] = cand[rng.integers(0, cand.size)]
    return seq


def _indel(rng, seq, p):
    """with probability p one insertion or deletion of 1..3 letters (a copy of the neighbouring letters half of the
    time: inside a repeat that is a slipped unit)"""
    if seq.size < 30 or rng.random() >= p:
        return seq
    at = int(rng.integers(10, seq.size - 10))
    k = int(rng.integers(1, 4))
    if rng.integers(0, 2):
        return np.delete(seq, slice(at, at + k))
    ins = seq[at:at + k].copy() if rng.integers(0, 2) else ACGT[rng.integers(0, 4, k)]
    return np.insert(seq, at, ins)


class Batch:
    """Host-side junction batch in the layout of include/dellyhip.h."""

    def __init__(self, chroms, junctions, seq_blob, seq_off, with_msa, truth):
        self.chroms = chroms            # list of np.uint8 arrays
        self.junctions = junctions      # structured array, abi.junction_dtype()
        self.seq_blob = seq_blob        # np.uint8
        self.seq_off = seq_off          # np.uint64, n_seq + 1
        self.with_msa = with_msa
        self.truth = truth              # list of dicts (generator ground truth)

    @property
    def n(self):
        return int(self.junctions.shape[0])

    @property
    def n_seq(self):
        return int(self.seq_off.shape[0] - 1)

    def seqs_of(self, i):
        j = self.junctions[i]
        first = int(j["seq_first"])
        return [bytes(self.seq_blob[int(self.seq_off[first + k]):int(self.seq_off[first + k + 1])])
                for k in range(int(j["n_seq"]))]


def subset(batch, idx):
    """The junctions idx (any order) of a batch as a batch of their own: same chromosomes, reads re-packed."""
    idx = np.asarray(idx, dtype=np.int64)
    junc = batch.junctions[idx].copy()
    parts, off, first = [], [0], 0
    for k, i in enumerate(idx):
        f, n = int(batch.junctions["seq_first"][i]), int(batch.junctions["n_seq"][i])
        for q in range(n):
            a, b = int(batch.seq_off[f + q]), int(batch.seq_off[f + q + 1])
            parts.append(batch.seq_blob[a:b])
            off.append(off[-1] + (b - a))
        junc["seq_first"][k] = first
        first += n
    blob = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    truth = [batch.truth[int(i)] for i in idx] if batch.truth is not None else None
    return Batch(batch.chroms, junc, blob, np.asarray(off, dtype=np.uint64), batch.with_msa, truth)


def _str(unit, n):
    u = np.frombuffer(unit, dtype=np.uint8)
    return np.tile(u, n // u.size + 2)[:n]


LOWCX_KINDS = 12


def plant_low_complexity(rng, G, s, e, sel, real=None):
    """Non-random sequence content around the breakpoints s (left) and e (right) of one private genome window -- the
    inputs on which the reference's tie-break rules decide the result (src/needle.h:107-123 join / refRight,
    :160-191 traceback order, src/gotoh.h:135-138, src/edlib.cpp:1021-1086, src/msa.h:46-89): homopolymers,
    short tandem repeats at and across the breakpoints, tandem duplications, a second copy of a flank inside the
    window, two-letter sequence, windows of a real chromosome (`real`, e.g. the reference's example/ref.fa).
    `rng` is a stream of its own, so the default batches do not change.  -> (G, kind name)"""
    W = G.size
    sel %= LOWCX_KINDS
    if real is not None and sel in (7, 8):
        o = int(rng.integers(0, real.size - W))
        G = real[o:o + W].copy()
    span = max(20, min(200, (e - s) // 2 if e - s > 60 else 200))
    if sel == 0:        # homopolymer across the left breakpoint
        L = int(rng.integers(20, span + 1)); a = int(rng.integers(0, L + 1))
        G[s - a:s - a + L] = ACGT[rng.integers(0, 4)]
        kind = "homopolymer"
    elif sel == 1:      # (CA)n across the left breakpoint
        L = int(rng.integers(20, span + 1)); a = int(rng.integers(0, L + 1))
        G[s - a:s - a + L] = _str(b"CA", L)
        kind = "str2"
    elif sel == 2:      # (CAG)n across the right breakpoint
        L = int(rng.integers(20, span + 1)); a = int(rng.integers(0, L + 1))
        G[e - a:e - a + L] = _str(b"CAG", L)
        kind = "str3"
    elif sel == 3:      # the same repeat at both breakpoints: long micro-homology, many co-optimal joins
        L = int(rng.integers(20, span + 1)); a = int(rng.integers(0, L + 1))
        unit = [b"CA", b"CAG", b"A", b"GATA"][int(rng.integers(0, 4))]
        ph = int(rng.integers(0, len(unit)))
        G[s - a:s - a + L] = _str(unit, L + ph)[ph:]
        G[e - a:e - a + L] = _str(unit, L)
        kind = "str_both"
    elif sel == 4:      # tandem duplication ending at the left breakpoint, unit 5 .. 100, 2 .. 4 copies
        u = int(rng.integers(5, 101)); c = int(rng.integers(2, 5))
        c = max(2, min(c, 300 // u))
        unit = G[s - u:s].copy()
        G[s - u * c:s] = np.tile(unit, c)
        if rng.integers(0, 2):
            G[e:e + u] = unit      # ... and once more behind the right breakpoint
        kind = "tandem"
    elif sel == 5:      # a second copy of the left flank behind the right breakpoint (and of the right flank before the left)
        L = int(rng.integers(100, 301))
        G[e + 20:e + 20 + L] = G[s - L:s]
        if rng.integers(0, 2):
            G[s - 40 - L:s - 40] = G[e:e + L]
        kind = "repeat"
    elif sel == 6:      # two-letter sequence
        lo, hi = max(0, s - 400), min(W, e + 400)
        G[lo:hi] = np.frombuffer(b"AT", dtype=np.uint8)[rng.integers(0, 2, hi - lo)]
        kind = "two_letter"
    elif sel == 7:
        kind = "real" if real is not None else "random"
    elif sel == 8:      # real sequence with a homopolymer at the left breakpoint
        L = int(rng.integers(10, 60))
        G[s - L // 2:s - L // 2 + L] = ord("T")
        kind = "real_homopolymer" if real is not None else "homopolymer"
    elif sel == 9:      # a deletion between two poly-A runs: the breakpoint can slide by the whole run
        L = int(rng.integers(10, 80)); R = int(rng.integers(10, 80))
        G[s - L:s + int(rng.integers(0, 20))] = ord("A")
        G[e - int(rng.integers(0, 20)):e + R] = ord("A")
        kind = "polya_both"
    elif sel == 10:     # repeats on the deleted side of both breakpoints
        L = int(rng.integers(20, span + 1))
        G[s:s + L] = _str(b"CA", L)
        G[e - L:e] = _str(b"CA", L)
        G[s - 10:s] = _str(b"CA", 10)
        kind = "str_inside"
    else:               # the whole neighbourhood is one periodic sequence (unit 2 .. 9) with a few point changes
        lo, hi = max(0, s - 450), min(W, e + 450)
        unit = ACGT[rng.integers(0, 4, int(rng.integers(2, 10)))]
        G[lo:hi] = np.tile(unit, (hi - lo) // unit.size + 1)[:hi - lo]
        G[rng.integers(lo, hi, max(4, (hi - lo) // 60))] = ACGT[rng.integers(0, 4, max(4, (hi - lo) // 60))]
        kind = "periodic"
    return G, kind


def make_batch(n, *, seed=42, first=0, mode="c2", n_reads=0, read_len=150, cons_flank=75,
               sub_rate=0.005, del_len=700, genome=None, real=None, read_indel=0.0, dup_reads=False, junction_ins=0):
    """Builds junctions first..first+n-1.

    mode "c2"   : BASELINE config 2 -- DEL, 150 bp consensus (or reads), ref
                  window 150+700+150 = 1000, plus 1 % each of N-run,
                  micro-homology and pure-reference (no split) junctions.
    mode "mixed": parity coverage -- DEL with l in 300..700 and 5000
                  (two-window branch src/split.h:117), DUP, INV 3to3/5to5 and
                  the four BND orientations across two chromosomes, varying
                  flank lengths.
    mode "allsvt": BASELINE config 3's mix -- "mixed" plus svt 4 insertions (one junction in 13), i.e. every SV type 0 .. 8
                  with the translocations on two chromosomes.
    mode "lr"   : long-read shapes (BASELINE config C4): ~2 kb consensus with 1 % substitutions
                  and indels, DEL 3 kb (window ~7 kb, contiguous branch), short DEL, DEL beyond
                  indelsize (two windows), INV, DUP; every third junction's consensus is given
                  reverse-complemented (the orientation test of src/split.h:564-572 must flip it).
                  Use with abi.params_lr(realign=True).
    mode "lrins": long-read insertions: svt 4, 800 bp of novel sequence (del_len = inserted
                  length), flanks of 1.2-1.5 kb, i.e. reads of ~3.8 kb; n_reads > 0: the loop body msaWfa +
                  alignConsensus (src/assemble.h:855-860).  Use with abi.params_lr(realign=True).
    mode "ins"  : svt 4 insertions (splitAlign path, src/split.h:480-538): 16..120 bp
                  novel or tandem-duplicated sequence, soft-masked / N-containing
                  reference stretches, pure-reference negatives.
    n_reads 0   : unit U (one consensus per junction); >0: unit U_full (that
                  many distinct split reads per junction, host order = as generated); a pair (lo, hi): that many,
                  drawn per junction from lo .. hi.
    genome      : None = uniform random letters; "lowcx" = plant_low_complexity() around the breakpoints of every junction
                  (kind = junction index mod LOWCX_KINDS; `real` = a real chromosome to cut windows from); "real" = every
                  window is cut from `real`.
    read_indel  : probability that a read / consensus gets one 1..3 bp indel (inside a repeat these are the co-optimal cases)
    dup_reads   : short reads only: every third read is repeated verbatim (the ABI takes what the host hands over; equal
                  distances everywhere are UPGMA's tie case, src/msa.h:46-89)
    junction_ins: that many non-templated bases between the two flanks of the ALT haplotype
    """
    two_chr = mode in ("mixed", "allsvt")
    nr_range = n_reads if isinstance(n_reads, (tuple, list)) else None
    if nr_range is not None:
        n_reads = int(nr_range[1])
    lr_like = mode in ("lr", "lrins")
    WINDOW = WINDOW_LR if lr_like else globals()["WINDOW"]
    chrA = np.empty(n * WINDOW, dtype=np.uint8)
    chrB = np.empty(n * WINDOW if two_chr else 0, dtype=np.uint8)
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs = []
    truth = []
    for k in range(n):
        j = first + k
        rng = _rng(seed, j)
        G = ACGT[rng.integers(0, 4, WINDOW)]
        base = k * WINDOW
        svt = 2
        s = 1500
        ell = del_len
        flankL = flankR = cons_flank
        kind = "del"
        H = None
        if two_chr:
            H = ACGT[rng.integers(0, 4, WINDOW)]
            sel = j % (13 if mode == "allsvt" else 12)
            flankL = int(rng.integers(40, 140))
            flankR = int(rng.integers(40, 140))
            if sel in (0, 1, 2, 3):
                ell = int(rng.integers(300, 701))
            elif sel == 4:
                ell = 1800  # > indelsize: two-window branch (kept inside the 4 kb window)
            elif sel == 5:
                kind, svt, ell = "dup", 3, int(rng.integers(200, 900))
            elif sel == 6:
                kind, svt, ell = "inv0", 0, int(rng.integers(400, 1500))
            elif sel == 7:
                kind, svt, ell = "inv1", 1, int(rng.integers(400, 1500))
            elif sel == 12:
                kind, svt = "ins", 4
                flankL = int(rng.integers(55, 98))
                flankR = int(rng.integers(55, 98))
                ell = int(rng.integers(20, 121))      # inserted length
            else:
                kind, svt = "bnd%d" % (sel - 8), 5 + (sel - 8)
        elif mode == "lr":
            s = 6000
            sel = j % 6
            flankL = int(rng.integers(700, 1150))
            flankR = int(rng.integers(700, 1150))
            ell = 3000
            if sel == 3:
                ell = int(rng.integers(400, 1500))
            elif sel == 4:
                kind, svt, ell = "inv0", 0, 3500
            elif sel == 5:
                kind, svt, ell = "dup", 3, 3000
        elif mode == "lrins":
            s = 6000
            svt = 4
            kind = "ins"
            flankL = int(rng.integers(1200, 1500))
            flankR = int(rng.integers(1200, 1500))
            ell = int(del_len) if del_len != 700 else 800
        elif mode == "ins":
            svt = 4
            sel = j % 10
            flankL = int(rng.integers(55, 98))    # consensus <= 97 + 120 + 97 = 314 (kernel limit 319)
            flankR = int(rng.integers(55, 98))
            ell = int(rng.integers(20, 121))      # inserted length
            kind = "ins"
            if sel == 6:
                kind = "insdup"                    # tandem duplication of the left flank -> homology
            elif sel == 7:
                kind = "insnone"                   # pure reference: splitAlign finds no gap
            elif sel == 8:
                kind = "insmask"                   # soft-masked reference + an N
            elif sel == 9:
                ell = int(rng.integers(14, 19))    # around the > 15 threshold of _validSRAlignment
        else:
            v = j % 100
            if v == 1:
                kind = "noref"       # pure reference: no split -> alignConsensus false
            elif v == 2:
                kind = "nrun"
            elif v == 3:
                kind = "hom"
        e = s + ell
        ins_seq = None
        if genome is not None:
            rng2 = _rng(seed ^ 0x10c0, j)
            if genome == "real":
                o = int(rng2.integers(0, real.size - WINDOW))
                G = real[o:o + WINDOW].copy()
                lkind = "real"
            else:
                G, lkind = plant_low_complexity(rng2, G, s, e if svt != 4 else s + 60, j, real)
                if H is not None and j % 2:
                    H, _ = plant_low_complexity(rng2, H, s, e, j, real)
        if svt == 4:
            e = s + int(rng.integers(0, 3))
            if kind == "insdup":
                ins_seq = G[s - ell:s].copy()
            elif kind == "insnone":
                ins_seq = G[0:0]
            else:
                ins_seq = ACGT[rng.integers(0, 4, ell)]
            if kind == "insmask":
                G[s - 40:s + 40] = G[s - 40:s + 40] + 32   # lower case
                G[s + 50] = ord("N")
        if kind == "nrun":
            G[s - 60:s - 50] = ord("N")
        if kind == "hom":
            h = int(rng.integers(2, 11))
            G[e:e + h] = G[s:s + h]
        # ALT haplotype around the junction (left part ‖ right part), 150+150
        L = max(1200 if mode == "lr" else 150, flankL, flankR) + (50 if mode == "lrins" else 0)
        if kind in ("del", "nrun", "hom"):
            left, right = G[s - L:s], G[e:e + L]
        elif kind == "noref":
            left, right = G[s - L:s], G[s:s + L]
        elif svt == 4:
            up = np.where((G >= 97) & (G <= 122), G - 32, G).astype(np.uint8)
            left, right = np.concatenate([up[s - L:s], ins_seq]), up[s:s + L]
            L = left.size
        elif kind == "dup":
            left, right = G[e - L:e], G[s:s + L]
        elif kind == "inv0":
            left, right = G[s - L:s], revcomp(G[e - L:e])
        elif kind == "inv1":
            left, right = revcomp(G[s:s + L]), G[e:e + L]
        elif kind == "bnd0":
            left, right = G[s - L:s], revcomp(H[e - L:e])
        elif kind == "bnd1":
            left, right = revcomp(G[s:s + L]), H[e:e + L]
        elif kind == "bnd2":
            left, right = G[s - L:s], H[e:e + L]
        elif kind == "bnd3":
            left, right = H[e - L:e], G[s:s + L]
        else:
            raise ValueError(kind)
        if junction_ins and svt != 4:
            left = np.concatenate([left, ACGT[_rng(seed ^ 0x1115, j).integers(0, 4, junction_ins)]])
            L = left.size
            flankL += junction_ins
        alt = np.concatenate([left, right])
        chrA[base:base + WINDOW] = G
        if two_chr:
            chrB[base:base + WINDOW] = H
        jit_s, jit_e = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
        rec = junc[k]
        rec["svid"] = j
        rec["svt"] = svt
        rec["chr"] = 0
        rec["chr2"] = 1 if kind.startswith("bnd") else 0
        rec["sv_start"] = base + s + jit_s
        rec["sv_end"] = base + e + jit_e
        rec["ins_len"] = 0
        if svt == 4:
            rec["sv_end"] = max(int(rec["sv_start"]), base + e + jit_e)
            rec["ins_len"] = max(0, ins_seq.size + int(rng.integers(-2, 3)))
            flankL += ins_seq.size
        rec["seq_first"] = len(seqs)
        if n_reads <= 0:
            cons = _mutate(rng, alt[L - flankL:L + flankR], sub_rate)
            if read_indel:
                cons = _indel(_rng(seed ^ 0x1de1, j), cons, read_indel)
            if lr_like:
                # sparse 1-base indels on top of the substitutions, then orientation
                keep = rng.random(cons.size) >= 0.004
                cons = cons[keep]
                at = np.nonzero(rng.random(cons.size) < 0.004)[0]
                cons = np.insert(cons, at, ACGT[rng.integers(0, 4, at.size)])
                if j % 3 == 2:
                    cons = revcomp(cons)
            seqs.append(cons)
            rec["n_seq"] = 1
        else:
            seen = set()
            want_reads = n_reads if nr_range is None else int(_rng(seed ^ 0x2ead, j).integers(int(nr_range[0]), int(nr_range[1]) + 1))
            lo, hi = 25, alt.size - read_len - 25
            if lr_like:   # long reads span the junction: ~L - 100 bases of each flank
                lo, hi = 0, 100
            tries = 0
            while len(seen) < want_reads and tries < 50 * want_reads:
                tries += 1
                o = int(rng.integers(lo, hi + 1))
                if lr_like:
                    r = _ont(rng, alt[o:alt.size - int(rng.integers(0, 101))], sub_rate)
                else:
                    r = _mutate(rng, alt[o:o + read_len], sub_rate)
                if read_indel:
                    r = _indel(_rng(seed ^ 0x1de1, j * 64 + tries), r, read_indel)
                key = r.tobytes()
                if key in seen:
                    continue
                seen.add(key)
                seqs.append(r)
                if dup_reads and len(seen) % 3 == 0 and len(seen) < want_reads:
                    seen.add(key + b"#%d" % len(seen))
                    seqs.append(r.copy())
            rec["n_seq"] = len(seen)
        if genome is not None:
            kind = kind + "/" + lkind
        truth.append(dict(kind=kind, svt=svt, start=base + s, end=base + e, flankL=flankL, flankR=flankR))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs], dtype=np.uint64)
    blob = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    chroms = [chrA, chrB] if two_chr else [chrA]
    return Batch(chroms, junc, blob, off, (2 if lr_like else 1) if n_reads > 0 else 0, truth)


def make_align_jobs(n_sv, reads_per_bp=40, *, seed=7, read_len=150, flank=13, sub_rate=0.005, weird=False):
    """Synthetic AlignJob batch for the split-read genotyping classifier (src/coverage.h:412-434).

    Per SV (a deletion) and breakpoint (bpPoint 0/1): consProbe / refProbe of length 2*flank + homLeft + homRight
    (src/coverage.h:231-256) cut from the ALT / REF haplotype around the breakpoint, and `reads_per_bp` reads of
    `read_len` bytes that span it -- half sampled from the ALT haplotype, half from the reference, with `sub_rate`
    substitutions, 2 % of them unrelated sequence ('N' results), qualities 0..60.  weird: longer probes
    (65 .. 256 bytes and beyond), ragged read lengths, lower-case / IUPAC bytes, empty strings.
    -> (jobs structured array, blob np.uint8)"""
    rng = np.random.default_rng(seed)
    parts = []
    pos = 0

    def put(a):
        nonlocal pos
        parts.append(a)
        o = pos
        pos += a.size
        return o

    rows = []
    for sv in range(n_sv):
        G = ACGT[rng.integers(0, 4, 1400)]
        s, e = 500, 900
        alt = np.concatenate([G[:s], G[e:]])
        for bp in (0, 1):
            hom = int(rng.integers(0, 12)) if not weird else int(rng.integers(0, 140))
            half = flank + hom // 2
            if weird and sv % 9 == 0:
                half = int(rng.integers(130, 170))      # beyond the 256-byte probe limit now and then
            centre_ref = s if bp == 0 else e
            consP = alt[s - half:s + half + (hom & 1)].copy()
            refP = G[centre_ref - half:centre_ref + half + (hom & 1)].copy()
            if weird and sv % 7 == 3:
                consP[rng.integers(0, consP.size)] = ord("R")
                refP[rng.integers(0, refP.size)] = ord("n")
            if weird and sv % 31 == 5:
                consP = consP[:0]
            co, ro = put(consP), put(refP)
            for r in range(reads_per_bp):
                L = read_len if not weird else int(rng.integers(0 if r % 13 == 0 else 30, 260))
                kind = r % 2
                if rng.random() < 0.02:
                    read = ACGT[rng.integers(0, 4, L)]
                else:
                    src, centre = (alt, s) if kind == 0 else (G, centre_ref)
                    o = centre - int(rng.integers(flank + 8, max(flank + 9, L - flank - 8))) if L > 2 * flank + 20 else centre - L // 2
                    o = max(0, o)
                    read = _mutate(rng, src[o:o + L].copy(), sub_rate)
                if weird and r % 11 == 4 and read.size:
                    read[rng.integers(0, read.size)] = rng.choice(np.frombuffer(b"NRYacgtn=", dtype=np.uint8))
                rows.append((co, ro, put(read), consP.size, refP.size, read.size, sv % 3, sv, int(rng.integers(0, 61))))
    jobs = np.zeros(len(rows), dtype=abi.align_job_dtype())
    for k, name in enumerate(("cons_off", "ref_off", "seq_off", "cons_len", "ref_len", "seq_len", "file_index", "sv_id", "qual")):
        jobs[name] = [r[k] for r in rows]
    blob = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    return jobs, blob


def make_nw_jobs(n_reads, *, seed=17, err=0.06, min_half=500, max_half=1000, weird=False):
    """Synthetic _editDistanceNW pairs of the long-read genotyper (src/genotype.h:262-284): per read and
    breakpoint the read slice `probe` (2*offset bytes, ONT-like errors) against the reference slice and against
    the consensus (ALT) slice of the same length; one of the two alleles is the read's own haplotype, the other
    differs by the SV (here a deletion next to the breakpoint), so one distance is small and one is large.
    weird: empty strings, very unequal lengths, bytes outside ACGT, pairs beyond the 6144-byte pattern limit.
    -> (jobs structured array, blob np.uint8)"""
    rng = np.random.default_rng(seed)
    parts, rows, pos = [], [], 0

    def put(a):
        nonlocal pos
        parts.append(a)
        o = pos
        pos += a.size
        return o

    for r in range(n_reads):
        half = int(rng.integers(min_half, max_half + 1))
        G = ACGT[rng.integers(0, 4, 2 * half + 800)]
        ref = G[:2 * half]
        alt = np.concatenate([G[:half], G[half + 700:half + 700 + half]])   # deletion of 700 bp at the breakpoint
        own = ref if r % 2 else alt
        probe = _ont(rng, own, err)[:2 * half]
        if weird:
            k = r % 9
            if k == 0:
                probe = probe[:0]
            elif k == 1:
                probe = probe[:int(rng.integers(1, 40))]
            elif k == 2 and probe.size:
                probe = probe.copy()
                probe[rng.integers(0, probe.size, 5)] = np.frombuffer(b"NRacg", dtype=np.uint8)
            elif k == 3:
                ref = np.concatenate([ref] * 4)[:6300 + int(rng.integers(0, 100))]       # pattern limit: the other string is shorter
            elif k == 4:
                ref = np.concatenate([ref] * 8)[:6200]
                probe = np.concatenate([probe] * 8)[:6150]                               # both beyond 6144: E_LIMIT
        po = put(np.ascontiguousarray(probe))
        rows.append((put(np.ascontiguousarray(ref)), po, ref.size, probe.size))
        rows.append((put(np.ascontiguousarray(alt)), po, alt.size, probe.size))
    jobs = np.zeros(len(rows), dtype=abi.nw_job_dtype())
    for k, name in enumerate(("query_off", "target_off", "query_len", "target_len")):
        jobs[name] = [x[k] for x in rows]
    return jobs, (np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8))


def load_real_chromosome(path=None):
    """The 2-bit packed real chromosome fixture (tests/golden/chr18_example.npz = the reference's example/ref.fa,
    made by tests/golden/make_real_fixture.py) as upper-case bytes."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "chr18_example.npz")
    z = np.load(path)
    p = z["packed"]
    c = np.stack([p & 3, (p >> 2) & 3, (p >> 4) & 3, p >> 6], axis=1).reshape(-1)[:int(z["n"])]
    return ACGT[c]


def tile_batch(batch, times):
    """`times` copies of a batch side by side on one (longer) genome: the chip-filling variants of bench.py's long-read rows are
    four tiles of the row's batch (generating 8 192 long-read junctions letter by letter takes minutes; the work per tile is
    the same, and tests/test_gpu_bench_shapes.py checks tile k against tile 0 and tile 0 against the reference)."""
    if times <= 1:
        return batch
    chroms = [np.tile(c, times) for c in batch.chroms]
    n = batch.n
    junc = np.tile(batch.junctions, times)
    nseq = batch.seq_off.size - 1
    for k in range(1, times):
        sl = slice(k * n, (k + 1) * n)
        junc["svid"][sl] += k * n
        junc["sv_start"][sl] += k * batch.chroms[0].size
        c2 = batch.junctions["chr2"]
        junc["sv_end"][sl] += np.where(c2 == 0, k * batch.chroms[0].size, k * batch.chroms[-1].size).astype(junc["sv_end"].dtype)
        junc["seq_first"][sl] += np.uint64(k * nseq)
    blob = np.tile(batch.seq_blob, times)
    total = int(batch.seq_off[-1])
    off = np.concatenate([batch.seq_off[:-1] + np.uint64(k * total) for k in range(times)] + [np.array([times * total], dtype=np.uint64)])
    truth = list(batch.truth) * times if batch.truth is not None else None
    return Batch(chroms, junc, blob, off, batch.with_msa, truth)


def make_big_deletions(shapes, seed=11, err=0.01, revcomp_every=0):
    """one DEL junction per (flank, ell): consensus = 2 * flank bases of the ALT haplotype with ONT-like errors -- BASELINE
    configs[3] as written is (5000, 700): a 10 kb consensus against a 20.7 kb reference window (longNeedle over 2 x 207 M cells;
    the reference holds four int32 matrices of 830 MB for it, src/needle.h:52-103).  Use with abi.params_lr(realign=True)."""
    rng = np.random.default_rng(seed)
    W = 70000
    n = len(shapes)
    chrom = ACGT[rng.integers(0, 4, n * W)]
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs = []
    for k, (flank, ell) in enumerate(shapes):
        s0 = k * W + 20000
        hap = np.concatenate([chrom[s0 - flank:s0], chrom[s0 + ell:s0 + ell + flank]])
        cons = _ont(rng, hap, err)
        if revcomp_every and k % revcomp_every == 1:
            cons = revcomp(cons)
        junc[k]["svid"] = k
        junc[k]["svt"] = 2
        junc[k]["sv_start"] = s0 + int(rng.integers(-3, 4))
        junc[k]["sv_end"] = s0 + ell + int(rng.integers(-3, 4))
        junc[k]["seq_first"] = k
        junc[k]["n_seq"] = 1
        seqs.append(cons)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    return Batch([chrom], junc, np.concatenate(seqs), off, 0, None)
