"""-m gpu parity tests of the long-read shapes (BASELINE config C4: ~2 kb consensus, ~7 kb
window, src/tegua.h:237-241 parameters, alignConsensus(..., realign=true)): the strip kernel
(delly_amd/csrc/lr_kernel.hpp) through the C-ABI against the C restatement and, when oracle/_ref
exists, the reference itself.  Integer / byte outputs: bit-exact."""
import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, INTERNAL_FOUND, compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr_ctx():
    ctx = refine.Context(params=abi.params_lr(realign=True))
    yield ctx
    ctx.close()


def test_lr_align_consensus_vs_port(lr_ctx, port):
    b = synth.make_batch(18, mode="lr", sub_rate=0.01)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL + INTERNAL_FOUND, label="hip-vs-port")
    assert int(gr["ok"].sum()) >= 15


def test_lr_vs_reference(lr_ctx, reference):
    b = synth.make_batch(6, mode="lr", sub_rate=0.01, first=100)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, rr, rb, label="hip-vs-reference")


def test_lr_without_realign_and_mixed_with_short(port):
    """realign off: reverse-complemented consensus sequences are NOT flipped (split.h:564);
    short-read-shaped junctions in the same batch still take the packed kernels"""
    ctx = refine.Context(params=abi.params_lr(realign=False))
    a = synth.make_batch(6, mode="lr", sub_rate=0.01, first=40)
    ctx.set_chromosomes(a.chroms)
    gr, gb = ctx.refine(a, want_alignment=False)
    pr, pb = port.refine_batch(a, params=abi.params_lr(realign=False), want_alignment=False)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="no-realign")
    ctx.close()


def test_lr_reproduces_reference_golden_vectors(lr_ctx):
    """HIP strip kernel vs the committed outputs of the reference itself (tests/golden/batch_u_lr.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_u_lr.npz"), allow_pickle=True)
    b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    compare(gr, gb, g["results"], g["blob"], label="batch_u_lr.npz")


def _lr_insertions(n, seed=3, flank=1200, ins=(200, 900), err=0.01, revcomp_every=0):
    """svt 4 junctions at long-read shapes with a given consensus (flank + inserted sequence + flank)"""
    rng = np.random.default_rng(seed)
    W = synth.WINDOW_LR
    chrom = synth.ACGT[rng.integers(0, 4, n * W)]
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs = []
    for k in range(n):
        s0 = k * W + 6000
        il = int(rng.integers(*ins))
        fl, fr = int(rng.integers(flank // 2, flank)), int(rng.integers(flank // 2, flank))
        hap = np.concatenate([chrom[s0 - fl:s0], synth.ACGT[rng.integers(0, 4, il)], chrom[s0:s0 + fr]])
        cons = synth._ont(rng, hap, err)
        if revcomp_every and k % revcomp_every == 1:
            cons = synth.revcomp(cons)
        junc[k]["svid"] = k
        junc[k]["svt"] = 4
        junc[k]["sv_start"] = s0 + int(rng.integers(-3, 4))
        junc[k]["sv_end"] = junc[k]["sv_start"] + 1
        junc[k]["ins_len"] = il
        junc[k]["seq_first"] = k
        junc[k]["n_seq"] = 1
        seqs.append(cons)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    return synth.Batch([chrom], junc, np.concatenate(seqs), off, 0, None)


def test_lr_insertions_vs_port(lr_ctx, port):
    """splitAlign with edlib in its Hirschberg regime (src/split.h:480-538 on ~2 kb strings), with the
    orientation test: every third consensus is given reverse-complemented"""
    b = _lr_insertions(9, revcomp_every=3)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL + INTERNAL_FOUND, label="hip-vs-port LR INS")
    assert int(gr["ok"].sum()) >= 7


def test_lr_insertions_vs_reference(lr_ctx, reference):
    b = _lr_insertions(4, seed=8, flank=900, ins=(300, 600), err=0.02)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, rr, rb, label="hip-vs-reference LR INS")
