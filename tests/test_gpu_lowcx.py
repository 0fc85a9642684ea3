"""-m gpu: tie-break parity on NON-random sequence.  Every kernel family against the reference itself (oracle/_ref) on
junctions whose genome is low-complexity or real: homopolymers, (CA)n / (CAG)n repeats at and across the breakpoints,
tandem duplications, a second copy of a flank inside the window, two-letter and periodic sequence, windows of the
reference's example chromosome (tests/golden/chr18_example.npz) -- synth.plant_low_complexity -- plus consensus /
reads with slipped-unit indels and unalignable ("junk") consensus sequences.  On such input the reference's
tie-break rules decide the result: join = first row-major maximum and LAST refRight (src/needle.h:107-123), the
no-gain test (:152), traceback vertical > horizontal > diagonal (:160-191), Gotoh's trace bits (src/gotoh.h:135-167),
edlib's INSERT > DELETE > diagonal and Hirschberg split (src/edlib.cpp:1021-1086,1328-1336), UPGMA's first row-major
maximum (src/msa.h:46-89).

Each batch runs with the sparse longNeedle kernels on and off (DELLYHIP_SR_SPARSE / DELLYHIP_SPARSE) and with and
without alignment rows (the two mask builders); CORE fields, consensus, alleles and alignment rows must be identical to
the reference's.  >= 2000 junctions per kernel family."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 128)
_REAL = None
_REF_CACHE = {}


def real():
    global _REAL
    if _REAL is None:
        _REAL = synth.load_real_chromosome()
    return _REAL


def _junk(batch, every=9, seed=3):
    """every `every`-th single-consensus junction gets a consensus that aligns nowhere (random letters)"""
    rng = np.random.default_rng(seed)
    blob = batch.seq_blob.copy()
    for j in range(4, batch.n, every):
        s = int(batch.junctions["seq_first"][j])
        a, b = int(batch.seq_off[s]), int(batch.seq_off[s + 1])
        blob[a:b] = synth.ACGT[rng.integers(0, 4, b - a)]
    return synth.Batch(batch.chroms, batch.junctions, blob, batch.seq_off, batch.with_msa, batch.truth)


def _env_refine(batch, env, want_alignment, params):
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


def _reference(reference, key, batch, params, threads=THREADS):
    if key not in _REF_CACHE:
        _REF_CACHE.clear()      # one batch at a time: alignment rows of long reads are large
        _REF_CACHE[key] = reference.refine_batch(batch, want_alignment=True, n_threads=threads, params=params)
    return _REF_CACHE[key]


def _check(reference, key, batch, params, variants, min_ok, threads=THREADS):
    rr, rb = _reference(reference, key, batch, params, threads)
    assert int(rr["ok"].sum()) >= min_ok, (key, int(rr["ok"].sum()))
    for env, want in variants:
        gr, gb = _env_refine(batch, env, want, params)
        fields = [f for f in CORE if want or f != "aln_len"]
        compare(gr, gb, rr, rb, fields=fields, blobs=("cons", "allele", "aln") if want else ("cons", "allele"),
                label="%s %s aln=%d" % (key, env, want))


SR_VARIANTS = [({"DELLYHIP_SR_SPARSE": "1"}, False), ({"DELLYHIP_SR_SPARSE": "1"}, True),
               ({"DELLYHIP_SR_SPARSE": "0"}, False), ({"DELLYHIP_SR_SPARSE": "0"}, True)]
LR_VARIANTS = [({"DELLYHIP_SPARSE": "1"}, False), ({"DELLYHIP_SPARSE": "1"}, True),
               ({"DELLYHIP_SPARSE": "0"}, False), ({"DELLYHIP_SPARSE": "0"}, True)]


@pytest.mark.parametrize("mode,seed", [("c2", 101), ("mixed", 102), ("c2", 103), ("mixed", 104)])
def test_sr_align_consensus_low_complexity(reference, mode, seed):
    """unit U (longNeedle: split_sparse_kernel / packed dense kernels / post kernel): 4 x 1200 junctions"""
    b = _junk(synth.make_batch(1200, seed=seed, mode=mode, genome="lowcx", real=real(), read_indel=0.3,
                               sub_rate=0.005 if seed < 103 else 0.012, junction_ins=0 if seed < 103 else 7))
    _check(reference, ("u", mode, seed), b, None, SR_VARIANTS, 600 if seed < 103 else 150)


def test_sr_align_consensus_real_windows(reference):
    b = synth.make_batch(2400, seed=105, mode="mixed", genome="real", real=real(), read_indel=0.2)
    _check(reference, ("u-real",), b, None, SR_VARIANTS, 1800)


@pytest.mark.parametrize("seed", [111, 112])
def test_sr_insertions_low_complexity(reference, seed):
    """svt 4 (splitAlign / edlib paths of ins_kernel): 2 x 1200 junctions"""
    b = synth.make_batch(1200, seed=seed, mode="ins", genome="lowcx", real=real())
    _check(reference, ("ins", seed), b, None, [({}, False), ({}, True)], 500)


@pytest.mark.parametrize("n_reads,n,dup,mode", [(5, 1200, False, "c2"), (20, 480, True, "c2"), (8, 480, True, "mixed"), (3, 480, False, "c2")])
def test_sr_msa_low_complexity(reference, n_reads, n, dup, mode):
    """unit U_full (LCS, UPGMA, profile Gotoh, consensus, then alignConsensus): 2640 junctions"""
    b = synth.make_batch(n, seed=120 + n_reads, mode=mode, n_reads=n_reads, genome="lowcx", real=real(), read_indel=0.25,
                         dup_reads=dup)
    _check(reference, ("ufull", n_reads, mode), b, None, [({"DELLYHIP_SR_SPARSE": "1"}, False), ({"DELLYHIP_SR_SPARSE": "0"}, True)], n // 2)


def test_sr_msa_identical_and_equidistant_reads(reference):
    """UPGMA ties (src/msa.h:46-89): error-free reads (many equal similarities) and verbatim repeats of a read"""
    b = synth.make_batch(600, seed=130, mode="c2", n_reads=9, sub_rate=0.0, dup_reads=True, genome="lowcx", real=real())
    _check(reference, ("ufull-ties",), b, None, [({}, True)], 300)


def test_lr_align_consensus_low_complexity(reference):
    """long-read alignConsensus(realign) (strip kernel: sparse tiles / dense strips, orientation test): 2 x 1008 junctions"""
    P = abi.params_lr(realign=True)
    for seed, rate in ((141, 0.01), (142, 0.002)):
        b = _junk(synth.make_batch(1008, seed=seed, mode="lr", sub_rate=rate, genome="lowcx", real=real(), read_indel=0.5), every=41)
        _check(reference, ("lr-u", seed), b, P, LR_VARIANTS if seed == 141 else LR_VARIANTS[:2], 700, threads=min(THREADS, 64))


@pytest.mark.parametrize("n_reads,n", [(15, 720), (6, 720), (3, 600)])
def test_lr_msa_edlib_low_complexity(reference, n_reads, n):
    """msaEdlib (all-pairs distances, medoid order, progressive NW PATH incl. Hirschberg, consensusEdlib) + alignConsensus(realign)"""
    P = abi.params_lr(realign=True)
    b = synth.make_batch(n, seed=150 + n_reads, mode="lr", n_reads=n_reads, sub_rate=0.06, genome="lowcx", real=real())
    _check(reference, ("lr-msaedlib", n_reads), b, P, [({"DELLYHIP_SPARSE": "1"}, False), ({"DELLYHIP_SPARSE": "0"}, True)], n // 2,
           threads=min(THREADS, 64))


@pytest.mark.parametrize("n_reads,n", [(15, 720), (6, 720), (3, 600)])
def test_lr_msa_wfa_insertions_low_complexity(reference, n_reads, n):
    """msaWfa (k-mer diagonals, superstring, progressive HW PATH, anchor trimming) + splitAlign in its Hirschberg regime"""
    P = abi.params_lr(realign=True)
    b = synth.make_batch(n, seed=160 + n_reads, mode="lrins", n_reads=n_reads, sub_rate=0.06, genome="lowcx", real=real())
    _check(reference, ("lr-msawfa", n_reads), b, P, [({}, False), ({}, True)], n // 3, threads=min(THREADS, 64))
