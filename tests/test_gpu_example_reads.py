"""-m gpu: BASELINE configs[0] as a parity fixture.  REAL reads from the reference's example BAMs (tests/golden/example_reads.npz,
made by tests/golden/make_example_reads.py with the test-only BAM reader tests/bamlite.py) against the real chromosome
(tests/golden/chr18_example.npz): the 8 kb deletion of the example data with its soft-clipped split reads (short reads:
msa + alignConsensus) and with slices of the ONT reads that support it (msaEdlib + alignConsensus(realign)), plus candidate
junctions at regular positions built from the reads that cover them (false candidates with real base errors), the long-read
ones also as insertion candidates (msaWfa + splitAlign).  Everything bit-compared with oracle/_ref."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _batch(name, with_msa):
    z = np.load(os.path.join(HERE, "golden", "example_reads.npz"))
    return synth.Batch([synth.load_real_chromosome()], z[name + "_junc"], z[name + "_blob"], z[name + "_off"], with_msa, None)


@pytest.mark.parametrize("want", [False, True])
def test_short_read_example_deletion_and_candidates(reference, want):
    b = _batch("sr", 1)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=want)
    ctx.close()
    rr, rb = reference.refine_batch(b, want_alignment=True)
    compare(gr, gb, rr, rb, fields=[f for f in CORE if want or f != "aln_len"], blobs=("cons", "allele", "aln") if want else ("cons", "allele"),
            label="example sr.bam")
    # the deletion the example data carries, at single-nucleotide resolution
    assert gr["ok"][0] == 1 and gr["sv_start"][0] == 100000 and gr["sv_end"][0] == 108002 and gr["sr_support"][0] == 10


@pytest.mark.parametrize("sparse", ["1", "0"])
def test_long_read_example_deletion_and_candidates(reference, sparse):
    P = abi.params_lr(realign=True)
    b = _batch("lr", 2)
    old = os.environ.get("DELLYHIP_SPARSE")
    os.environ["DELLYHIP_SPARSE"] = sparse
    try:
        ctx = refine.Context(params=P)
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=True)
        ctx.close()
    finally:
        if old is None:
            os.environ.pop("DELLYHIP_SPARSE", None)
        else:
            os.environ["DELLYHIP_SPARSE"] = old
    rr, rb = reference.refine_batch(b, want_alignment=True, params=P, n_threads=min(os.cpu_count() or 1, 16))
    compare(gr, gb, rr, rb, fields=CORE, label="example lr.bam")
    assert gr["ok"][0] == 1 and gr["sv_start"][0] == 100000 and gr["sv_end"][0] == 108001


def test_long_read_slices_as_insertion_candidates(reference):
    P = abi.params_lr(realign=True)
    b = _batch("lrins", 2)
    ctx = refine.Context(params=P)
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=True)
    ctx.close()
    rr, rb = reference.refine_batch(b, want_alignment=True, params=P, n_threads=min(os.cpu_count() or 1, 16))
    compare(gr, gb, rr, rb, fields=CORE, label="example lr.bam as insertions")
    assert (gr["cons_len"] > 1500).all()      # msaWfa built a consensus for every candidate


def test_single_item_wrappers_on_real_reads(reference, gpu_ctx):
    """msa / msaEdlib / msaWfa through the single-item entry points on the example deletion's read sets"""
    sr, lr = _batch("sr", 1), _batch("lr", 2)
    reads = sr.seqs_of(0)
    assert gpu_ctx.msa(reads) == reference.msa(reads)
    slices = lr.seqs_of(0)
    assert gpu_ctx.msa_edlib(slices) == reference.msa_edlib(slices)
    for k in (1, 2):
        s = lr.seqs_of(k)
        assert gpu_ctx.msa_edlib(s) == reference.msa_edlib(s)
        assert gpu_ctx.msa_wfa(s) == reference.msa_wfa(s)


@pytest.mark.parametrize("sparse", ["1", "0"])
def test_real_split_reads_as_dup_inv_bnd_on_rearranged_real_sequence(reference, sparse, monkeypatch):
    """The example data only carries a deletion: the other six SV types get the SAME real split reads on chromosomes made of
    rearranged real chr18 sequence (tests/golden/make_example_reads.py, `svx`): DUP, INV 3to3 / 5to5, BND 3to5 / 5to3 / 3to3 /
    5to5 -- the reference refines every one to the base -- plus reads of the wrong strand and ordinary reads as candidates of
    every type.  msa() + alignConsensus(), sparse kernel on and off, against oracle/_ref."""
    z = np.load(os.path.join(HERE, "golden", "example_reads.npz"))
    chroms = [z["svx_chr%d" % i] for i in range(int(z["svx_nchr"]))]
    b = synth.Batch(chroms, z["svx_junc"], z["svx_blob"], z["svx_off"], 1, None)
    monkeypatch.setenv("DELLYHIP_SR_SPARSE", sparse)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=True)
    ctx.close()
    rr, rb = reference.refine_batch(b, want_alignment=True)
    compare(gr, gb, rr, rb, fields=CORE, label="rearranged example reads")
    svt = b.junctions["svt"][:7].tolist()
    assert svt == [3, 0, 1, 7, 8, 5, 6] and gr["ok"][:7].tolist() == [1] * 7 and int(gr["ok"][7:].sum()) == 0
    assert (gr["sr_align_quality"][:7] == 1.0).all() and (gr["sr_support"][:7] == 10).all()
