"""-m gpu parity tests of the split-alignment path (unit U), through the C-ABI:
HIP kernels vs the C restatement (and vs the reference itself when oracle/_ref
was built) on the same seeded batches.  Integer/byte outputs: bit-exact."""
import numpy as np
import pytest

from delly_amd import synth
from util import CORE, INTERNAL, INTERNAL_FOUND, compare

pytestmark = pytest.mark.gpu


def _run(gpu_ctx, batch, want_alignment=True):
    gpu_ctx.set_chromosomes(batch.chroms)
    return gpu_ctx.refine(batch, want_alignment=want_alignment)


def test_dpp_lane_shift_semantics(gpu_ctx):
    # the DP hand-off relies on wave_shr:1 / wave_shl:1 crossing all 64 lanes;
    # a wrong shift would break every alignment below, so one tiny case first
    found, r0, r1 = gpu_ctx.long_needle(b"ACGTACGTTTGACCAGTACGATCGATTTGACA" * 2, b"ACGTACGTTTGACCAGTACGATCGATTTGACA" + b"G" * 40 + b"ACGTACGTTTGACCAGTACGATCGATTTGACA")
    assert found
    assert len(r0) == len(r1)


@pytest.mark.parametrize("mode,n", [("c2", 400), ("mixed", 360), ("ins", 400)])
def test_align_consensus_vs_port(gpu_ctx, port, mode, n):
    b = synth.make_batch(n, mode=mode)
    gr, gb = _run(gpu_ctx, b)
    pr, pb = port.refine_batch(b)
    # (for svt 4 the INTERNAL_FOUND slots carry splitAlign's csStart/csEnd/bestJoin/leftEnd/rightStart)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL + (INTERNAL_FOUND if mode == "ins" else []), label="hip-vs-port")
    assert int(gr["ok"].sum()) > (0.7 if mode == "ins" else 0.9) * n


def test_insertions_mixed_with_other_types(gpu_ctx, port):
    """svt 4 junctions are routed to their own kernel: a batch that interleaves them with
    deletions must give the same records as the two homogeneous batches"""
    a = synth.make_batch(64, mode="ins", seed=5)
    gpu_ctx.set_chromosomes(a.chroms)
    ra, _ = gpu_ctx.refine(a, want_alignment=False)
    pa, _ = port.refine_batch(a, want_alignment=False)
    for f in CORE:
        assert np.array_equal(ra[f], pa[f]), f
    # same junction records inside a larger deletion batch sharing the chromosome
    import numpy.lib.recfunctions as rfn  # noqa: F401
    d = synth.make_batch(64, mode="c2", seed=5)
    chrom = np.concatenate([a.chroms[0], d.chroms[0]])
    jd = d.junctions.copy()
    jd["sv_start"] += a.chroms[0].size
    jd["sv_end"] += a.chroms[0].size
    jd["seq_first"] += a.n_seq
    junc = np.concatenate([a.junctions, jd])
    perm = np.random.default_rng(3).permutation(junc.shape[0])
    off = np.concatenate([a.seq_off, d.seq_off[1:] + a.seq_off[-1]])
    mixed = synth.Batch([chrom], junc[perm], np.concatenate([a.seq_blob, d.seq_blob]), off, False, None)
    gpu_ctx.set_chromosomes(mixed.chroms)
    rm, _ = gpu_ctx.refine(mixed, want_alignment=False)
    pm, _ = port.refine_batch(mixed, want_alignment=False)
    for f in CORE:
        assert np.array_equal(rm[f], pm[f]), f


def test_align_consensus_vs_reference(gpu_ctx, reference):
    b = synth.make_batch(240, mode="mixed", first=1000)
    gr, gb = _run(gpu_ctx, b)
    rr, rb = reference.refine_batch(b)
    compare(gr, gb, rr, rb, label="hip-vs-reference")


def test_long_needle_single(gpu_ctx, port):
    rng = np.random.default_rng(7)
    for it in range(40):
        m = int(rng.integers(20, 300))
        n = int(rng.integers(60, 1500))
        ref = rng.choice(list(b"ACGT"), n).astype(np.uint8)
        # consensus = two pieces of ref + noise, sometimes unrelated
        if it % 5 == 4:
            cons = rng.choice(list(b"ACGT"), m).astype(np.uint8)
        else:
            a = int(rng.integers(0, max(1, n // 2 - m // 2)))
            bpos = int(rng.integers(n // 2, max(n // 2 + 1, n - m // 2)))
            cons = np.concatenate([ref[a:a + m // 2], ref[bpos:bpos + (m - m // 2)]])
            cons = cons.copy()
            for k in rng.integers(0, cons.size, max(1, cons.size // 50)):
                cons[k] = rng.choice(list(b"ACGT"))
        f1, a0, a1 = gpu_ctx.long_needle(cons.tobytes(), ref.tobytes())
        f2, b0, b1, _ = port.long_needle(cons.tobytes(), ref.tobytes())
        assert f1 == f2, (it, m, n)
        assert a0 == b0 and a1 == b1, (it, m, n)


def test_edge_characters(gpu_ctx, port):
    # lower case, N runs and IUPAC letters in consensus and reference
    rng = np.random.default_rng(11)
    for it in range(30):
        n = int(rng.integers(200, 900))
        ref = rng.choice(list(b"ACGTNRYacgtn"), n, p=[.2, .2, .2, .2, .05, .02, .02, .03, .03, .03, .01, .01]).astype(np.uint8)
        m = int(rng.integers(40, 200))
        a = int(rng.integers(0, n // 3))
        bpos = int(rng.integers(n // 2, n - m // 2 - 1))
        cons = np.concatenate([ref[a:a + m // 2], ref[bpos:bpos + (m - m // 2)]]).copy()
        f1, a0, a1 = gpu_ctx.long_needle(cons.tobytes(), ref.tobytes())
        f2, b0, b1, _ = port.long_needle(cons.tobytes(), ref.tobytes())
        assert (f1, a0, a1) == (f2, b0, b1), it


def test_hip_reproduces_reference_golden_vectors(gpu_ctx):
    """HIP path vs the committed outputs of the reference itself (tests/golden)."""
    import glob
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    n_checked = 0
    for path in sorted(glob.glob(os.path.join(gold, "batch_u_*.npz"))):
        g = np.load(path, allow_pickle=True)
        if "lr" in g.files and int(g["lr"]):
            continue  # long-read parameters: tests/test_gpu_lr.py
        b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
        gr, gb = _run(gpu_ctx, b)
        compare(gr, gb, g["results"], g["blob"], label=os.path.basename(path))
        n_checked += b.n
    assert n_checked >= 200
    g = np.load(os.path.join(gold, "primitives.npz"), allow_pickle=True)
    for s1, s2, f, r0, r1 in zip(g["ln_s1"], g["ln_s2"], g["ln_found"], g["ln_r0"], g["ln_r1"]):
        hf, h0, h1 = gpu_ctx.long_needle(s1, s2)
        assert (int(hf), h0, h1) == (int(f), r0, r1)


def test_full_size_properties(gpu_ctx):
    """BASELINE config-2 size (10 000 junctions): size-independent properties --
    every planted deletion is recovered at single-nucleotide resolution up to
    its micro-homology, results do not depend on batch composition/order."""
    n = 10000
    b = synth.make_batch(n, mode="c2")
    gpu_ctx.set_chromosomes(b.chroms)
    res, _ = gpu_ctx.refine(b, want_alignment=False)
    kinds = np.array([t["kind"] for t in b.truth])
    start = np.array([t["start"] for t in b.truth])
    end = np.array([t["end"] for t in b.truth])
    dele = kinds != "noref"
    assert int(res["ok"][dele].sum()) >= int(0.995 * dele.sum())
    assert int(res["ok"][~dele].sum()) == 0
    okd = dele & (res["ok"] == 1)
    # reported breakpoints bracket the truth within the homology wiggle (+1 for the 1-based end)
    assert np.all(np.abs(res["sv_start"][okd] - start[okd]) <= res["ci_wiggle"][okd] + 1)
    assert np.all(np.abs(res["sv_end"][okd] - end[okd]) <= res["ci_wiggle"][okd] + 1)
    # idempotence / order independence: a permuted sub-batch gives identical records
    sub = synth.make_batch(512, mode="c2", first=4096)
    gpu_ctx.set_chromosomes(sub.chroms)
    r2, _ = gpu_ctx.refine(sub, want_alignment=False)
    base = 4096 * synth.WINDOW
    for f in ("ok", "ci_wiggle", "hom_len", "cons_bp", "sr_align_quality", "matches", "mismatches"):
        assert np.array_equal(r2[f], res[f][4096:4096 + 512]), f
    assert np.array_equal(r2["sv_start"] + base, res["sv_start"][4096:4096 + 512])


def test_empty_batch_and_size_boundaries(gpu_ctx, port):
    """n = 0; consensus of exactly 319 (last short-read shape) and 320..420 bp (routed to the strip
    kernel with the short-read parameters); consensus shorter than 2*minimumFlankSize."""
    b0 = synth.make_batch(0, mode="c2")
    gpu_ctx.set_chromosomes(synth.make_batch(1, mode="c2").chroms)
    r0, bl0 = gpu_ctx.align_consensus_batch(b0.junctions, np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert r0.shape[0] == 0 and bl0.size == 0
    for flank in (159, 160, 165, 210):   # m = 2*flank: 318, 320, 330, 420  (+ one batch at 319 below)
        b = synth.make_batch(24, mode="c2", cons_flank=flank, seed=9)
        gr, gb = _run(gpu_ctx, b)
        pr, pb = port.refine_batch(b)
        compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="flank=%d" % flank)
        assert int(gr["ok"].sum()) >= 20
    b = synth.make_batch(12, mode="c2", cons_flank=160, seed=10)
    # trim one base: m = 319
    seqs = [b.seqs_of(i)[0][:319] for i in range(b.n)]
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    b319 = synth.Batch(b.chroms, b.junctions, np.frombuffer(b"".join(seqs), dtype=np.uint8), off, False, b.truth)
    gr, gb = _run(gpu_ctx, b319)
    pr, pb = port.refine_batch(b319)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="m=319")
    assert set(gr["cons_len"].tolist()) == {319}
    short = synth.make_batch(8, mode="c2", cons_flank=10)   # 20 bp < 2*13: alignConsensus false (split.h:647)
    gr, gb = _run(gpu_ctx, short)
    pr, pb = port.refine_batch(short)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL)
    assert int(gr["ok"].sum()) == 0
