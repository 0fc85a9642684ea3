"""-m gpu: shapes beyond the round-1 kernel limits, bit-compared with the reference itself (oracle/_ref):
  * longNeedle / alignConsensus with consensus > 4159 and window > 24000 (BASELINE configs[3]: 10 kb x 20 kb),
    incl. the orientation test on patterns beyond one bit-vector pass (6144 rows);
  * msaEdlib and msaWfa on reads > 6144 bytes (src/assemble.h:807-816 slices reads to +-(1000 + CI + inslen));
  * _editDistanceNW pairs with both strings > 6144 bytes."""
import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr_ctx():
    ctx = refine.Context(params=abi.params_lr(realign=True))
    yield ctx
    ctx.close()


@pytest.fixture
def lr_reference(reference):
    """the session-wide reference checker with the `delly lr` parameters for the single-item calls, restored afterwards"""
    old = reference.params
    reference.params = abi.params_lr()
    yield reference
    reference.params = old


_big_deletions = synth.make_big_deletions   # (shared with bench.py's lr_stress_10kb_x_20kb row)


def test_long_needle_10kb_x_20kb_and_beyond_vs_reference(lr_ctx, reference):
    # (flank, DEL length): m ~ 2*flank; contiguous window n = 2m + ell while ell <= indelsize (10 000), else two windows
    shapes = [(2500, 9000),     # m 5 000, n 19 000: beyond the old 4159-row limit, one bit-vector pass
              (3300, 7000),     # m 6 600 > 6144: the orientation test runs in strips; n 20 200
              (5000, 10001),    # m 10 000, two windows clipped at the midpoint: n ~ 30 000 (> the old 24 000)
              (5000, 700)]      # m 10 000, n 20 700: BASELINE's 10 kb x 20 kb
    b = _big_deletions(shapes, revcomp_every=2)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True), n_threads=4)
    compare(gr, gb, rr, rb, label="big longNeedle")
    assert int(gr["ok"].sum()) == len(shapes)
    assert int((gr["status"] != 0).sum()) == 0
    assert gr["cons_len"].max() > 9000 and gr["ref_len"].max() > 25000


def _reads(rng, n, length, err, ins=0):
    base = synth.ACGT[rng.integers(0, 4, length + 100)]
    if ins:
        mid = base.size // 2
        base = np.concatenate([base[:mid], synth.ACGT[rng.integers(0, 4, ins)], base[mid:]])
    L = base.size
    return [bytes(synth._ont(rng, base[int(rng.integers(0, 40)):L - int(rng.integers(0, 40))], err)) for _ in range(n)], base


def test_msa_edlib_reads_beyond_6144_vs_reference(lr_ctx, lr_reference):
    reference = lr_reference
    rng = np.random.default_rng(21)
    for n, length in ((4, 7000), (6, 9000), (3, 12000)):
        reads, _ = _reads(rng, n, length, 0.05)
        want = reference.msa_edlib(reads)
        got = lr_ctx.msa_edlib(reads)
        assert got == want, (n, length, got[0], want[0], len(got[1]), len(want[1]))
        assert len(got[1]) > 6144


def test_msa_wfa_reads_beyond_6144_vs_reference(lr_ctx, lr_reference):
    reference = lr_reference
    rng = np.random.default_rng(22)
    for n, length, ins in ((4, 5000, 2500), (5, 6000, 4000)):
        reads, base = _reads(rng, n, length, 0.04, ins=ins)
        assert max(len(r) for r in reads) > 6144
        pre, suf = bytes(base[:1000]), bytes(base[-1000:])
        for anchors in ((b"", b""), (pre, suf)):
            want = reference.msa_wfa(reads, *anchors)
            got = lr_ctx.msa_wfa(reads, *anchors)
            assert got == want, (n, length, ins, bool(anchors[0]), got[0], want[0], len(got[1]), len(want[1]))


def test_refine_batch_lr_insertion_with_long_reads_vs_reference(lr_ctx, reference):
    """the long-read loop body for a 3 kb insertion: msaWfa on ~8 kb read slices + alignConsensus (splitAlign)"""
    rng = np.random.default_rng(23)
    W = 40000
    n = 3
    chrom = synth.ACGT[rng.integers(0, 4, n * W)]
    junc = np.zeros(n, dtype=abi.junction_dtype())
    seqs, first = [], 0
    for k in range(n):
        s0 = k * W + 15000
        il = int(rng.integers(2500, 3500))
        ins = synth.ACGT[rng.integers(0, 4, il)]
        hap = np.concatenate([chrom[s0 - 2600:s0], ins, chrom[s0:s0 + 2600]])
        nr = 5
        for _ in range(nr):
            a, e = int(rng.integers(0, 60)), hap.size - int(rng.integers(0, 60))
            seqs.append(synth._ont(rng, hap[a:e], 0.04))
        junc[k]["svid"] = k
        junc[k]["svt"] = 4
        junc[k]["sv_start"] = s0 + int(rng.integers(-2, 3))
        junc[k]["sv_end"] = junc[k]["sv_start"] + 1
        junc[k]["ins_len"] = il
        junc[k]["seq_first"] = first
        junc[k]["n_seq"] = nr
        first += nr
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    b = synth.Batch([chrom], junc, np.concatenate(seqs), off, 2, None)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=False)
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True), want_alignment=False, n_threads=3)
    compare(gr, gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label="LR INS long reads")
    assert int((gr["status"] != 0).sum()) == 0
    assert int(gr["ok"].sum()) >= 2


def test_edit_distance_nw_pairs_beyond_6144_vs_reference(gpu_ctx, reference):
    rng = np.random.default_rng(24)
    strs = []
    for L in (6200, 9000, 13000, 20000):
        a = synth.ACGT[rng.integers(0, 4, L)]
        strs.append((a, synth._ont(rng, a, 0.06)))
    strs.append((strs[1][0], strs[3][1]))   # unrelated strings of different length
    blob = np.concatenate([x for p in strs for x in p])
    jobs = np.zeros(len(strs), dtype=abi.nw_job_dtype())
    o = 0
    for i, (a, c) in enumerate(strs):
        jobs[i]["query_off"], jobs[i]["query_len"] = o, a.size
        o += a.size
        jobs[i]["target_off"], jobs[i]["target_len"] = o, c.size
        o += c.size
    got = gpu_ctx.edit_distance_nw_batch(jobs, blob)
    want = reference.edit_distance_nw_batch(jobs, blob)
    assert (got == want).all(), (got, want)
    assert (got > 0).all()
