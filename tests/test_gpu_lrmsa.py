"""-m gpu parity tests of the long-read MSA (msaEdlib, src/assemble.h:383-473) and of the
long-read loop body msaEdlib + alignConsensus(realign) through the C-ABI, against vectors of the
reference itself (tests/golden/longread.npz) and the C restatement.  Byte outputs: bit-exact."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lr_ctx():
    ctx = refine.Context(params=abi.params_lr(realign=True))
    yield ctx
    ctx.close()


def test_msa_edlib_reproduces_reference_golden_vectors(lr_ctx):
    g = np.load(os.path.join(GOLD, "longread.npz"), allow_pickle=True)
    for reads, rows, cs in zip(g["msa_sets"], g["msa_rows"], g["msa_cs"]):
        r, c = lr_ctx.msa_edlib(list(reads))
        assert r == int(rows)
        assert c == cs, (len(c), len(cs))


def _ont(rng, s, rate):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(rng.choice(list(b"ACGT")))
            out.append(ch)
            continue
        if u < rate:
            out.append(rng.choice(list(b"ACGT")))
            continue
        out.append(ch)
    return bytes(out)


def test_msa_edlib_vs_port_small_and_edge(lr_ctx, port):
    rng = np.random.default_rng(31)
    old = port.params
    port.params = abi.params_lr()
    try:
        for it in range(8):
            L = int(rng.integers(120, 900))
            base = bytes(rng.choice(list(b"ACGT"), L + 60).astype(np.uint8))
            n = [1, 2, 3, 4, 7, 12, 15, 16][it]
            reads = [_ont(rng, base[int(rng.integers(0, 30)):L + 30 + int(rng.integers(0, 30))], 0.08) for _ in range(n)]
            if it % 3 == 1:   # letters outside the 15 equality classes (N, lower case): the byte-compare variant of the pass
                reads = [bytes((ord("N") if (k % 41 == 7) else (c + 32 if k % 53 == 11 else c)) for k, c in enumerate(r)) for r in reads]
            assert lr_ctx.msa_edlib(reads) == port.msa_edlib(reads), (it, n, L)
    finally:
        port.params = old


def _subs(rng, s, rate):
    out = bytearray(s)
    for k in range(len(out)):
        if rng.random() < rate:
            out[k] = rng.choice(list(b"ACGT"))
    return bytes(out)


def test_msa_edlib_wide_row_blocks(lr_ctx, port):
    """shapes that take the two- and three-word variants of the bit-vector passes: a short read against a consensus of
    > 2048 columns (traceback-regime rectangle with > 2048 rows), consensus of > 4096 columns (Hirschberg halves > 2048 rows)"""
    rng = np.random.default_rng(77)
    old = port.params
    port.params = abi.params_lr()
    try:
        base = bytes(rng.choice(list(b"ACGT"), 4400).astype(np.uint8))
        # (msaEdlib drops the worst 20 % of the reads: two short ones so that one of them stays)
        sets = [[_ont(rng, base[:2180], 0.03), _ont(rng, base[10:2190], 0.03), _ont(rng, base[400:1800], 0.03), _ont(rng, base[:2170], 0.03),
                 _ont(rng, base[300:1750], 0.03)],
                # substitutions only: the alignment stays at 4100 columns (kernel limit 4160), halves of 2050 rows
                [_subs(rng, base[:4100], 0.01), _subs(rng, base[:4100], 0.01), _subs(rng, base[1000:3300], 0.02), _subs(rng, base[:4100], 0.01),
                 _subs(rng, base[900:3200], 0.02)],
                # very short reads against 4100 columns: a traceback-regime rectangle with > 4096 rows (three words per lane)
                [_subs(rng, base[:4100], 0.01), _subs(rng, base[:4100], 0.01), _subs(rng, base[2000:2300], 0.02), _subs(rng, base[:4100], 0.01),
                 _subs(rng, base[1500:1810], 0.02)]]
        for k, reads in enumerate(sets):
            want = port.msa_edlib(reads)
            assert lr_ctx.msa_edlib(reads) == want, k
    finally:
        port.params = old


def test_refine_batch_lr_reproduces_reference_golden_vectors(lr_ctx):
    """msaEdlib + alignConsensus(realign) vs the committed outputs of the reference (batch_full_lr_n8.npz)"""
    g = np.load(os.path.join(GOLD, "batch_full_lr_n8.npz"), allow_pickle=True)
    b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
    assert b.with_msa == 2
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    compare(gr, gb, g["results"], g["blob"], label="batch_full_lr_n8.npz")
    assert int(gr["ok"].sum()) == b.n


def test_refine_batch_lr_vs_port(lr_ctx, port):
    b = synth.make_batch(6, mode="lr", n_reads=5, sub_rate=0.05, seed=77, first=20)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=False)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True), want_alignment=False)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="hip-vs-port")


def test_msa_wfa_reproduces_reference_golden_vectors(lr_ctx):
    """msaWfa (src/assemble.h:547-726) vs vectors produced by the reference (with and without anchors)"""
    g = np.load(os.path.join(GOLD, "longread.npz"), allow_pickle=True)
    n = 0
    for reads, pre, suf, rows, cs in zip(g["wfa_sets"], g["wfa_pre"], g["wfa_suf"], g["wfa_rows"], g["wfa_cs"]):
        r, c = lr_ctx.msa_wfa(list(reads), pre, suf)
        assert r == int(rows), n
        assert c == cs, (n, len(c), len(cs))
        n += 1
    assert n >= 5


def test_msa_wfa_vs_port(lr_ctx, port):
    rng = np.random.default_rng(41)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    old = port.params
    port.params = abi.params_lr()
    try:
        for it in range(6):
            F = int(rng.integers(200, 900))
            left = bytes(rng.choice(list(b"ACGT"), F + 400).astype(np.uint8))
            right = bytes(rng.choice(list(b"ACGT"), F + 400).astype(np.uint8))
            hap = left + bytes(rng.choice(list(b"ACGT"), int(rng.integers(60, 500))).astype(np.uint8)) + right
            n = [2, 3, 5, 8, 12, 16][it]
            reads = []
            for k in range(n):
                a = 400 - int(rng.integers(0, min(F, 350)))
                b = len(hap) - 400 + int(rng.integers(0, min(F, 350)))
                r = _ont(rng, hap[a:b], 0.07 if it % 2 else 0.02)
                if it == 3 and k % 4 == 1:
                    r = bytes(comp[c] for c in reversed(r))
                reads.append(r)
            pre, suf = (left[-250:], right[:250]) if it % 2 == 0 else (b"", b"")
            assert lr_ctx.msa_wfa(reads, pre, suf) == port.msa_wfa(reads, pre, suf), (it, n)
    finally:
        port.params = old


def _lr_insertion_reads(n, n_reads=6, seed=5, err=0.04):
    rng = np.random.default_rng(seed)
    W = synth.WINDOW_LR
    chrom = synth.ACGT[rng.integers(0, 4, n * W)]
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs = []
    for k in range(n):
        s0 = k * W + 6000
        il = int(rng.integers(300, 800))
        hap = np.concatenate([chrom[s0 - 1300:s0], synth.ACGT[rng.integers(0, 4, il)], chrom[s0:s0 + 1300]])
        junc[k]["svid"] = k
        junc[k]["svt"] = 4
        junc[k]["sv_start"] = s0
        junc[k]["sv_end"] = s0 + 1
        junc[k]["ins_len"] = il
        junc[k]["seq_first"] = len(seqs)
        junc[k]["n_seq"] = n_reads
        for _ in range(n_reads):
            seqs.append(synth._ont(rng, hap[int(rng.integers(0, 100)):hap.size - int(rng.integers(0, 100))], err))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    return synth.Batch([chrom], junc, np.concatenate(seqs), off, 2, None)


def test_refine_batch_lr_insertions_vs_port(lr_ctx, port):
    """long-read insertion branch of the loop body: msaWfa with reference anchors + alignConsensus(realign=false)"""
    b = _lr_insertion_reads(4)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="hip-vs-port LR INS loop")
    assert int(gr["ok"].sum()) >= 3


def test_refine_batch_lr_small_inversions(lr_ctx, port):
    """src/assemble.h:840-853: for INV junctions smaller than the consensus only the middle svSize letters
    are aligned, the consensus is restored afterwards and consBp shifted"""
    rng = np.random.default_rng(12)
    n = 4
    W = synth.WINDOW_LR
    chrom = synth.ACGT[rng.integers(0, 4, n * W)]
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs = []
    for k in range(n):
        s0 = k * W + 6000
        L = int(rng.integers(900, 1500))
        hap = np.concatenate([chrom[s0 - 1300:s0], synth.revcomp(chrom[s0:s0 + L]), chrom[s0 + L:s0 + L + 1300]])
        junc[k]["svid"] = k
        junc[k]["svt"] = k % 2
        junc[k]["sv_start"] = s0
        junc[k]["sv_end"] = s0 + L
        junc[k]["seq_first"] = len(seqs)
        junc[k]["n_seq"] = 5
        for _ in range(5):
            seqs.append(synth._ont(rng, hap[int(rng.integers(0, 100)):hap.size - int(rng.integers(0, 100))], 0.04))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    b = synth.Batch([chrom], junc, np.concatenate(seqs), off, 2, None)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=False)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True), want_alignment=False)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="small inversions")
    assert np.all(gr["cons_len"] > 3000)   # the restored, untrimmed consensus is reported
