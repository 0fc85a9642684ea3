"""-m gpu: the sparse longNeedle kernels (CHANGELOG.md 3.0 / 3.7) against the dense kernels they stand in front of and against
the oracle: same records and bytes whether a junction is finished by split_sparse_kernel, left to the packed dense
kernels (letters outside ACGTN, lower case, deficits beyond the level budget, windows beyond the tile) or forced
through the dense path (DELLYHIP_SR_SPARSE=0 / DELLYHIP_SPARSE=0), with and without alignment rows (the two mask
builders of split_sparse.hpp)."""
import os

import numpy as np
import pytest

import fuzz
from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu


def _fields(want_alignment):
    """the oracle always returns the alignment rows; without want_alignment the product leaves aln_len = 0"""
    return [f for f in CORE + INTERNAL if want_alignment or f != "aln_len"]


def _refine(batch, env, want_alignment, params=None):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = refine.Context(params=params)   # (the knobs are read at dellyhip_create)
        ctx.set_chromosomes(batch.chroms)
        out = ctx.refine(batch, want_alignment=want_alignment)
        ctx.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def _noisy(batch, rate, seed, lower=0.0, n_rate=0.0):
    """substitutions (and optionally lower-case / N letters) in every second consensus of a c2 batch"""
    rng = np.random.default_rng(seed)
    blob = batch.seq_blob.copy()
    for j in range(0, batch.n, 2):
        s = int(batch.junctions["seq_first"][j])
        a, b = int(batch.seq_off[s]), int(batch.seq_off[s + 1])
        seg = blob[a:b]
        hit = rng.random(seg.size) < rate
        seg[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(hit.sum()))]
        if lower:
            lo = rng.random(seg.size) < lower
            seg[lo] |= 0x20
        if n_rate:
            seg[rng.random(seg.size) < n_rate] = ord("N")
    return synth.Batch(batch.chroms, batch.junctions, blob, batch.seq_off, batch.with_msa, batch.truth)


@pytest.mark.parametrize("want_alignment", [False, True])
@pytest.mark.parametrize("mode", ["c2", "mixed"])
def test_sparse_vs_dense_vs_oracle(port, mode, want_alignment):
    b = fuzz.perturbed(400, 31, mode)
    gs, bs = _refine(b, {"DELLYHIP_SR_SPARSE": "1"}, want_alignment)
    gd, bd = _refine(b, {"DELLYHIP_SR_SPARSE": "0"}, want_alignment)
    blobs = ("cons", "allele", "aln") if want_alignment else ("cons", "allele")
    compare(gs, bs, gd, bd, fields=CORE + INTERNAL, blobs=blobs, label="sparse vs dense " + mode)
    pr, pb = port.refine_batch(b)
    assert int(pr["ok"].sum()) > 150
    compare(gs, bs, pr, pb, fields=_fields(want_alignment), blobs=blobs, label="sparse vs oracle " + mode)


@pytest.mark.parametrize("rate,lower,n_rate", [(0.02, 0.0, 0.0), (0.08, 0.0, 0.0), (0.2, 0.0, 0.0), (0.01, 0.05, 0.0), (0.01, 0.0, 0.03)])
def test_levels_budget_and_unclean_letters(port, rate, lower, n_rate):
    """2 % .. 20 % substitutions walk the level schedule 0, 2, .., 32 and past it (dense fallback); lower-case and N
    letters in the consensus take the exact byte-wise path"""
    b = _noisy(synth.make_batch(600, mode="c2", seed=5), rate, 17, lower, n_rate)
    for want in (False, True):
        gs, bs = _refine(b, {"DELLYHIP_SR_SPARSE": "1"}, want)
        pr, pb = port.refine_batch(b)
        compare(gs, bs, pr, pb, fields=_fields(want), blobs=("cons", "allele", "aln") if want else ("cons", "allele"),
                label="noisy %.2f/%.2f/%.2f" % (rate, lower, n_rate))


def test_long_read_sparse_vs_dense_strips():
    P = abi.params_lr(realign=True)
    b = synth.make_batch(96, mode="lr", sub_rate=0.01)
    gs, bs = _refine(b, {"DELLYHIP_SPARSE": "1", "DELLYHIP_SPARSE_COST": "400"}, False, P)
    gd, bd = _refine(b, {"DELLYHIP_SPARSE": "0"}, False, P)
    assert int(gd["ok"].sum()) > 60
    compare(gs, bs, gd, bd, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="lr sparse vs dense")


def test_unsplit_score_is_known_and_exact_when_there_is_no_split(port):
    """needle.h:152: a consensus that aligns without a split (the false-positive candidate) makes longNeedle return false
    BECAUSE bestScore == mat[m][n] -- on the sparse path that value comes from the first level whose furthest row reaches m,
    so for these junctions score_unsplit must be reported (not DELLYHIP_SCORE_UNKNOWN) and equal the dense matrices' value.
    (A pure-reference consensus with a substitution near one end can still split at a smaller deficit than it aligns
    whole -- a three-base tail matching somewhere downstream; longNeedle returns true there, the later filters reject it,
    and score_unsplit may stay unknown: the level that reaches row m was never needed.)"""
    from util import SCORE_UNKNOWN
    b = synth.make_batch(3000, mode="c2", seed=9)
    noref = np.array([t["kind"] == "noref" for t in b.truth])
    sub = synth.subset(b, np.nonzero(noref)[0])
    assert sub.n >= 25
    sub = _noisy(sub, 0.01, 3)                      # (most of them with a few substitutions: deficit > 0)
    gs, bs = _refine(sub, {"DELLYHIP_SR_SPARSE": "1"}, False)
    pr, pb = port.refine_batch(sub)
    assert (pr["ok"] == 0).all() and (gs["ok"] == 0).all()
    nosplit = pr["score_best"] == pr["score_unsplit"]          # longNeedle returned false (needle.h:152)
    assert int(nosplit.sum()) >= sub.n // 2 and int((pr["score_unsplit"][nosplit] < pr["cons_len"][nosplit]).sum()) >= 5
    assert (gs["score_unsplit"][nosplit] == pr["score_unsplit"][nosplit]).all()
    known = gs["score_unsplit"] != SCORE_UNKNOWN
    assert (gs["score_unsplit"][known] == pr["score_unsplit"][known]).all()
    assert (gs["score_best"] == pr["score_best"]).all()
    compare(gs, bs, pr, pb, fields=_fields(False), blobs=("cons", "allele"), label="no-split junctions")
