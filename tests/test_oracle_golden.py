"""CPU: the C restatement (oracle/liboracle.so) against the golden vectors that
tests/golden/make_golden.py produced from the reference's own headers
(oracle/_ref), and -- when oracle/_ref is present -- against the reference
directly on fresh seeds.  Integer / byte outputs: bit-exact."""
import glob
import os

import numpy as np
import pytest

import pyoracle
from delly_amd import abi, synth
from util import CORE, compare, compare_probes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _batches():
    return sorted(glob.glob(os.path.join(GOLD, "batch_*.npz")))


@pytest.mark.parametrize("path", _batches(), ids=lambda p: os.path.basename(p))
def test_port_reproduces_golden_batches(port, path):
    g = np.load(path, allow_pickle=True)
    kw = eval(str(g["kwargs"]))
    b = synth.make_batch(int(g["n"]), **kw)
    lr = "lr" in g.files and int(g["lr"])
    res, blob = port.refine_batch(b, want_alignment=True, params=abi.params_lr(realign=True) if lr else None)
    compare(res, blob, g["results"], g["blob"], label=os.path.basename(path))


def test_port_reproduces_golden_primitives(port):
    g = np.load(os.path.join(GOLD, "primitives.npz"), allow_pickle=True)
    for a, b, o in zip(g["lcs_a"], g["lcs_b"], g["lcs_out"]):
        assert port.lcs(a, b) == o
    for x, y in zip(g["rc_in"], g["rc_out"]):
        assert port.reverse_complement(x) == y
    for a, b, o in zip(g["hom_a"], g["hom_b"], g["hom_out"]):
        assert port.longest_homology(a, b) == o
    for s1, s2, f, r0, r1 in zip(g["ln_s1"], g["ln_s2"], g["ln_found"], g["ln_r0"], g["ln_r1"]):
        pf, p0, p1, _ = port.long_needle(s1, s2)
        assert (int(pf), p0, p1) == (int(f), r0, r1)
    for reads, rows, cs, root, d, p in zip(g["msa_sets"], g["msa_rows"], g["msa_cs"], g["tree_root"], g["tree_d"],
                                           g["tree_p"]):
        reads = list(reads)
        assert port.msa(reads) == (int(rows), cs)
        r2, d2, p2 = port.guide_tree(reads)
        assert r2 == root and np.array_equal(d2, d) and np.array_equal(p2, p)
    for a1, a2, sc, rows, cons in zip(g["gotoh_a1"], g["gotoh_a2"], g["gotoh_score"], g["gotoh_rows"],
                                      g["gotoh_cons"]):
        s2, r2 = port.gotoh(list(a1), list(a2))
        assert s2 == sc and r2 == list(rows)
        assert port.consensus(list(rows)) == cons


def test_port_reproduces_golden_edlib(port):
    """the plain-DP restatement of edlib against vectors produced by the reference's vendored edlib"""
    g = np.load(os.path.join(GOLD, "edlib.npz"), allow_pickle=True)
    for q, t, mode, out, ops in zip(g["q"], g["t"], g["mode"], g["out"], g["ops"]):
        r = port.edlib_align(q, t, int(mode), 2)
        assert r is not None
        assert tuple(r[:4]) == tuple(int(x) for x in out) and r[4] == ops, (len(q), len(t), int(mode))
    for cons, ref, rc, r0, r1 in zip(g["sa_cons"], g["sa_ref"], g["sa_rc"], g["sa_row0"], g["sa_row1"]):
        prc, p0, p1, _ = port.split_align(cons, ref)
        assert (prc, p0, p1) == (int(rc), r0, r1)


def test_port_reproduces_golden_long_read_vectors(port):
    """edlib's Hirschberg regime, the extended-IUPAC equalities, and msaEdlib (src/assemble.h:383-473)"""
    g = np.load(os.path.join(GOLD, "longread.npz"), allow_pickle=True)
    for q, t, mode, out, ops in zip(g["q"], g["t"], g["mode"], g["out"], g["ops"]):
        r = port.edlib_align(q, t, int(mode), 2)
        assert tuple(r[:4]) == tuple(int(x) for x in out) and r[4] == ops, (len(q), len(t), int(mode))
    old = port.params
    port.params = abi.params_lr()
    try:
        for reads, rows, cs in zip(g["msa_sets"], g["msa_rows"], g["msa_cs"]):
            assert port.msa_edlib(list(reads)) == (int(rows), cs)
        for reads, pre, suf, rows, cs in zip(g["wfa_sets"], g["wfa_pre"], g["wfa_suf"], g["wfa_rows"], g["wfa_cs"]):
            assert port.msa_wfa(list(reads), pre, suf) == (int(rows), cs)
    finally:
        port.params = old


def test_port_edlib_vs_reference_fresh(port, reference):
    rng = np.random.default_rng(99)
    for it in range(300):
        t = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 330))).astype(np.uint8))
        if it % 3 == 0:
            q = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 260))).astype(np.uint8))
        else:
            a = int(rng.integers(0, len(t)))
            q = bytearray(t[a:a + int(rng.integers(1, 200))])
            for k in range(len(q)):
                if rng.random() < 0.05:
                    q[k] = rng.choice(list(b"ACGT"))
            q = bytes(q)
        for mode in (0, 1, 2):
            assert port.edlib_align(q, t, mode) == reference.edlib_align(q, t, mode), (it, mode)


@pytest.mark.parametrize("mode,n_reads,n", [("c2", 0, 150), ("mixed", 0, 180), ("mixed", 7, 36), ("c2", 12, 12),
                                            ("ins", 0, 150), ("ins", 6, 24)])
def test_port_vs_reference_fresh_seeds(port, reference, mode, n_reads, n):
    b = synth.make_batch(n, mode=mode, n_reads=n_reads, seed=777, first=5000)
    rr, rb = reference.refine_batch(b)
    pr, pb = port.refine_batch(b)
    compare(pr, pb, rr, rb, label="port-vs-reference")


@pytest.mark.parametrize("genome", ["lowcx", "real"])
@pytest.mark.parametrize("mode,n_reads,n,extra", [("c2", 0, 240, {}), ("mixed", 0, 120, dict(junction_ins=6)), ("mixed", 5, 36, dict(dup_reads=True)),
                                                  ("ins", 0, 60, {})])
def test_port_vs_reference_where_the_tie_breaks_decide(port, reference, genome, mode, n_reads, n, extra):
    """the restatement against the reference's own code on the sequence of the round-3 GPU suite (tests/test_gpu_lowcx.py):
    homopolymers, STRs, tandem duplications, repeated segments (every low-complexity kind twenty times at n = 240), and windows
    of the real chromosome -- join, refRight, traceback and UPGMA ties (src/needle.h:107-123,160-191, src/msa.h:46-89)"""
    real = synth.load_real_chromosome() if genome == "real" else None
    b = synth.make_batch(n, mode=mode, n_reads=n_reads, seed=4242, genome=genome, real=real, **extra)
    rr, rb = reference.refine_batch(b)
    pr, pb = port.refine_batch(b)
    compare(pr, pb, rr, rb, label="port-vs-reference %s %s" % (genome, mode))
    assert int(rr["ok"].sum()) >= n // 4


@pytest.mark.parametrize("kw", [dict(mode="lr", sub_rate=0.02, seed=171), dict(mode="lr", n_reads=5, sub_rate=0.06, seed=172),
                                dict(mode="lrins", n_reads=5, sub_rate=0.06, seed=173)])
def test_port_vs_reference_long_read_low_complexity(port, reference, kw):
    """the long-read loop bodies (msaEdlib / msaWfa + alignConsensus) on low-complexity windows: edlib's co-optimal paths
    (src/edlib.cpp:1021-1086) and the progressive consensus on repeats"""
    b = synth.make_batch(8, genome="lowcx", real=synth.load_real_chromosome(), **kw)
    p = abi.params_lr(realign=True)
    pr, pb = port.refine_batch(b, params=p, n_threads=8)
    rr, rb = reference.refine_batch(b, params=p, n_threads=8)
    compare(pr, pb, rr, rb, label="lr lowcx")
    assert int(rr["ok"].sum()) >= 3


@pytest.mark.parametrize("kw", [dict(mode="lr", n_reads=5, sub_rate=0.05, seed=91), dict(mode="lr", sub_rate=0.02, seed=92),
                                dict(mode="lr", n_reads=4, sub_rate=0.03, seed=93, first=4)])
def test_port_vs_reference_long_read_fresh(port, reference, kw):
    """the long-read loop body (msaEdlib / given consensus + alignConsensus(realign)) on batches not in the fixtures"""
    b = synth.make_batch(6, **kw)
    p = abi.params_lr(realign=True)
    pr, pb = port.refine_batch(b, params=p, n_threads=6)
    rr, rb = reference.refine_batch(b, params=p, n_threads=6)
    compare(pr, pb, rr, rb, label="lr fresh")
    assert int(rr["ok"].sum()) >= 4


def test_port_reproduces_golden_align_jobs(port):
    """split-read genotyping classifier (src/coverage.h:412-434): reference-generated records"""
    z = np.load(os.path.join(GOLD, "align_jobs.npz"))
    for label, fq in (("plain", 0.95), ("weird", 0.95), ("lowq", 0.4)):
        p = abi.params_sr()
        p.flank_quality = fq
        got = port.classify_reads(z[label + "_jobs"], z[label + "_blob"], params=p)
        assert got.tobytes() == z[label + "_results"].tobytes(), label


def test_port_reproduces_golden_nw_jobs(port):
    """long-read genotyping (src/genotype.h:21-30): reference-generated distances"""
    z = np.load(os.path.join(GOLD, "nw_jobs.npz"))
    for label in ("plain", "weird"):
        got = port.edit_distance_nw_batch(z[label + "_jobs"], z[label + "_blob"], n_threads=4)
        assert (got == z[label + "_dist"]).all(), label


def test_port_reproduces_golden_probes(port):
    """per-SV body of _generateProbes (src/coverage.h:196-258): reference-generated probes"""
    z = np.load(os.path.join(GOLD, "probes.npz"))
    for label in ("c2", "mixed", "ins"):
        b = synth.make_batch(int(z[label + "_n"]), **eval(str(z[label + "_kwargs"])))
        rec, blob = port.generate_probes(b)
        compare_probes(rec, blob, z[label + "_rec"], z[label + "_blob"], label)


def test_port_probes_vs_reference_fuzz(port, reference):
    import fuzz
    for mode in ("c2", "mixed", "ins"):
        for pi in (0, 1, 3):
            b = fuzz.perturbed(120, 9 + pi, mode)
            p = fuzz.params_of(pi)
            a, ab = port.generate_probes(b, params=p)
            r, rb = reference.generate_probes(b, params=p)
            compare_probes(a, ab, r, rb, "%s/%d" % (mode, pi))
        for b in fuzz.clipped(20, 5, mode):
            a, ab = port.generate_probes(b)
            r, rb = reference.generate_probes(b)
            compare_probes(a, ab, r, rb, "clipped " + mode)


def test_port_classifier_vs_reference_fresh(port, reference):
    for seed, weird, fq in ((21, False, 0.95), (22, True, 0.95), (23, True, 0.45), (24, True, 0.0)):
        jobs, blob = synth.make_align_jobs(25, 12, seed=seed, weird=weird)
        p = abi.params_sr()
        p.flank_quality = fq
        a = port.classify_reads(jobs, blob, params=p, n_threads=3)
        b = reference.classify_reads(jobs, blob, params=p, n_threads=2)
        assert a.tobytes() == b.tobytes(), (seed, fq)


@pytest.mark.parametrize("mode", ["c2", "mixed", "ins"])
def test_fuzz_port_vs_reference(port, reference, mode):
    """perturbed breakpoint estimates / consensus sequences under unusual parameter sets, and reference windows
    clipped at the chromosome ends (tests/fuzz.py); the same inputs go through the HIP path in tests/test_gpu_fuzz.py"""
    import fuzz
    for pi in range(4):
        p = fuzz.params_of(pi)
        b = fuzz.perturbed(240, 5 + pi, mode)
        rr, rb = reference.refine_batch(b, params=p)
        pr, pb = port.refine_batch(b, params=p)
        compare(pr, pb, rr, rb, label="fuzz %s/%d" % (mode, pi))
    for b in fuzz.clipped(60, 3, mode):
        rr, rb = reference.refine_batch(b)
        pr, pb = port.refine_batch(b)
        compare(pr, pb, rr, rb, label="clipped " + mode)


def test_port_multithreaded_matches_single(port):
    b = synth.make_batch(64, mode="mixed", seed=5)
    r1, b1 = port.refine_batch(b, n_threads=1, want_alignment=False)
    r4, b4 = port.refine_batch(b, n_threads=4, want_alignment=False)
    compare(r1, b1, r4, b4, blobs=("cons", "allele"), label="threads")


def test_edge_cases(port, reference):
    """too-short consensus (split.h:647), single read (shortpe.h:166), N runs."""
    b = synth.make_batch(8, mode="c2", cons_flank=10)   # 20 bp consensus < 2*13
    rr, rb = reference.refine_batch(b)
    pr, pb = port.refine_batch(b)
    compare(pr, pb, rr, rb)
    assert int(rr["ok"].sum()) == 0
    b = synth.make_batch(6, mode="c2", n_reads=1)
    rr, rb = reference.refine_batch(b)
    pr, pb = port.refine_batch(b)
    compare(pr, pb, rr, rb)
    assert int(rr["sr_support"].sum()) == 0


def test_port_long_read_insertion_loop_vs_reference(port, reference):
    """src/assemble.h:855-860: msaWfa with reference anchors, then alignConsensus(realign=false) whose
    splitAlign runs edlib in its Hirschberg regime"""
    rng = np.random.default_rng(1)
    b0 = synth.make_batch(2, mode="lr", n_reads=5, sub_rate=0.04)
    junc = b0.junctions.copy()
    seqs = []
    for k in range(2):
        G = b0.chroms[0][k * synth.WINDOW_LR:(k + 1) * synth.WINDOW_LR]
        s0 = 6000
        hap = np.concatenate([G[s0 - 1200:s0], synth.ACGT[rng.integers(0, 4, 500)], G[s0:s0 + 1200]])
        junc[k]["svt"] = 4
        junc[k]["sv_start"] = k * synth.WINDOW_LR + s0
        junc[k]["sv_end"] = k * synth.WINDOW_LR + s0 + 1
        junc[k]["ins_len"] = 500
        junc[k]["seq_first"] = len(seqs)
        junc[k]["n_seq"] = 5
        for _ in range(5):
            seqs.append(synth._ont(rng, hap[int(rng.integers(0, 100)):hap.size - int(rng.integers(0, 100))], 0.04))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    b = synth.Batch(b0.chroms, junc, np.concatenate(seqs), off, 2, None)
    pl = abi.params_lr(realign=True)
    rr, rb = reference.refine_batch(b, params=pl)
    pr, pb = port.refine_batch(b, params=pl)
    compare(pr, pb, rr, rb, label="port-vs-reference LR INS")
    assert int(rr["ok"].sum()) == 2


def test_c_restatement_under_address_and_undefined_behaviour_sanitizers():
    """SURVEY.md 5 (sanitizers): oracle/delly_oracle.c compiled with -fsanitize=address,undefined and driven over seeded junction
    batches (msa + alignConsensus, given-consensus path, primitives at their edges) by oracle/sanitize_selftest.c"""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and ("cannot find -lasan" in r.stdout or "libasan" in r.stdout and "No such file" in r.stdout):
        import pytest
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stdout[-3000:]
    assert "sanitize_selftest: done rc=0" in r.stdout
