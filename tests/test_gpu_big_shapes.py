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


def _split_pair(rng, m, n, err=0.01, kind="del"):
    """a consensus of m letters spanning a deletion (or carrying an insertion) against a window of n letters"""
    ref = synth.ACGT[rng.integers(0, 4, n)]
    if kind == "del":
        a = int(rng.integers(50, max(51, n // 2 - m // 2)))
        b = int(rng.integers(n // 2 + m // 2, max(n // 2 + m // 2 + 1, n - m // 2 - 50)))
        cons = np.concatenate([ref[a:a + m // 2], ref[b:b + (m - m // 2)]]).copy()
    else:
        ins = max(30, m // 4)
        a = int(rng.integers(10, max(11, n - (m - ins) - 10)))
        half = (m - ins) // 2
        cons = np.concatenate([ref[a:a + half], synth.ACGT[rng.integers(0, 4, ins)], ref[a + half:a + (m - ins)]]).copy()
    for k in rng.integers(0, cons.size, max(1, int(err * cons.size))):
        cons[k] = synth.ACGT[rng.integers(0, 4)]
    return cons.tobytes(), ref.tobytes()


def test_single_item_long_needle_at_any_shape_vs_reference(gpu_ctx, lr_reference):
    """dellyhip_long_needle beyond 319 x 2 048 runs the long-read kernels in their direct mode (the reference has no
    limit, src/needle.h:45-47) -- incl. the 2 kb x 7 kb shape of the long-read loop and an unrelated pair (longNeedle returns false)"""
    rng = np.random.default_rng(5)
    for it, (m, n) in enumerate([(2000, 7000), (400, 1500), (300, 2500), (1500, 1800), (3000, 9000)]):
        cons, ref = _split_pair(rng, m, n)
        if it == 3:
            cons = synth.ACGT[rng.integers(0, 4, m)].tobytes()
        f1, a0, a1 = gpu_ctx.long_needle(cons, ref)
        f2, b0, b1, _ = lr_reference.long_needle(cons, ref)
        assert f1 == f2, (m, n)
        assert a0 == b0 and a1 == b1, (m, n)


def test_single_item_split_align_at_any_shape_vs_reference(gpu_ctx, lr_reference):
    """dellyhip_split_align beyond the short-read shapes (src/split.h:480-482 has no limit): lr_ins_kernel's direct mode"""
    rng = np.random.default_rng(6)
    for m, n in [(1800, 2400), (900, 3000), (2500, 2600)]:
        cons, ref = _split_pair(rng, m, n, kind="ins")
        f1, a0, a1 = gpu_ctx.split_align(cons, ref)
        f2, b0, b1 = lr_reference.split_align(cons, ref)[:3]
        assert f1 == bool(f2), (m, n)
        if f1:
            assert a0 == b0 and a1 == b1, (m, n)
