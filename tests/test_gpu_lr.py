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


def _unalignable(b, every=1, seed=5):
    """the batch with the consensus of every `every`-th junction replaced by random letters of the same length: the sparse
    longNeedle gives up on those after its first rounds and the dense strips run"""
    rng = np.random.default_rng(seed)
    blob = b.seq_blob.copy()
    for k in range(0, b.n, every):
        o = int(b.junctions["seq_first"][k])
        lo, hi = int(b.seq_off[o]), int(b.seq_off[o + 1])
        blob[lo:hi] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, hi - lo)]
    return synth.Batch(b.chroms, b.junctions, blob, b.seq_off, b.with_msa, b.truth)


def _refine_with_teams(b, teams, monkeypatch, serial=False):
    monkeypatch.setenv("DELLYHIP_LR_TEAMS", str(teams))
    if serial:
        monkeypatch.setenv("DELLYHIP_LR_TEAMS_SERIAL", "1")
    else:
        monkeypatch.delenv("DELLYHIP_LR_TEAMS_SERIAL", raising=False)
    ctx = refine.Context(params=abi.params_lr(realign=True))
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync()
    rb.run(); rb.sync()          # (the second run finds the counters and the list of the first)
    stats = rb.lr_team_stats()
    res0, blob0 = rb.fetch()
    rb.free()
    res, blob = ctx.refine(b, want_alignment=True)   # (the one-shot entry point: with the alignment rows)
    compare(res0, blob0, res, blob, fields=[f for f in CORE + INTERNAL + INTERNAL_FOUND if f != "aln_len"], blobs=("cons", "allele"), label="resident vs one-shot")
    ctx.close()
    return res, blob, stats


def test_dense_strips_on_teams_vs_reference(monkeypatch, reference):
    """lr_dense_team_kernel (CHANGELOG.md 3.7): junctions whose consensus does not align go to teams of four wavefronts that sweep
    the strips of the dense longNeedle pipelined.  Same records as the reference, and the teams really took them."""
    b = _unalignable(synth.make_batch(12, mode="lr", sub_rate=0.01, first=300), every=2)
    gr, gb, stats = _refine_with_teams(b, 64, monkeypatch)
    assert stats[0] == 12 and stats[1] >= 6 and stats[3] == 0, stats
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, rr, rb, label="teams-vs-reference")
    assert int(gr["ok"].sum()) <= 6


def test_teams_equal_single_wavefront_strips(monkeypatch):
    """the same batch with the teams beside lr_kernel, after it on one stream, with fewer teams than dense junctions (the list
    is taken in turns; beyond two junctions per team lr_kernel sweeps the strips itself) and without teams: bit-identical"""
    b = _unalignable(synth.make_batch(96, mode="lr", sub_rate=0.01, first=700), every=3)
    base, base_blob, s0 = _refine_with_teams(b, 0, monkeypatch)
    assert s0 == (0, 0, 0, 0)
    for teams, serial in ((64, False), (64, True), (3, False), (1, False)):
        res, blob, stats = _refine_with_teams(b, teams, monkeypatch, serial=serial)
        label = "teams %d serial %s" % (teams, serial)
        assert stats[0] == min(teams, b.n) and stats[3] == 0, (label, stats)
        assert stats[1] == min(2 * stats[0], stats[1]) and stats[1] >= min(2 * teams, 30), (label, stats)
        compare(res, blob, base, base_blob, fields=CORE + INTERNAL + INTERNAL_FOUND, label=label)
