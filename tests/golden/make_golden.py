#!/usr/bin/env python
"""Generates tests/golden/*.npz from THE REFERENCE ITSELF (oracle/_ref: the
reference's own headers compiled against oracle/shim).  The reference ships no
tests or golden vectors, so these seeded inputs + reference
outputs are the pinned vectors for the path.  Runs only where /root/reference
exists; the .npz files are committed and travel.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pyoracle  # noqa: E402
from delly_amd import abi, synth  # noqa: E402

BATCHES = {
    # name: (n, kwargs)
    "u_c2": (96, dict(mode="c2", seed=42, first=0)),
    "u_mixed": (120, dict(mode="mixed", seed=42, first=0)),
    "full_c2_n8": (24, dict(mode="c2", seed=43, first=0, n_reads=8)),
    "full_mixed_n5": (36, dict(mode="mixed", seed=44, first=0, n_reads=5)),
    "full_c2_n20": (6, dict(mode="c2", seed=45, first=0, n_reads=20)),
    "u_ins": (160, dict(mode="ins", seed=46, first=0)),
    "full_ins_n6": (30, dict(mode="ins", seed=47, first=0, n_reads=6)),
    # long-read shapes (BASELINE config C4) with `delly lr` parameters and realign=true
    "u_lr": (10, dict(mode="lr", seed=48, first=0, sub_rate=0.01)),
    # long-read loop body: msaEdlib over 8 ONT-like reads (6 % errors) + alignConsensus(realign)
    "full_lr_n8": (4, dict(mode="lr", seed=49, first=0, n_reads=8, sub_rate=0.06)),
}
LR_BATCHES = {"u_lr", "full_lr_n8"}


def random_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(list(alphabet), n).astype(np.uint8))


def edlib_vectors(ref):
    """edlibAlign (NW/SHW/HW, PATH) and splitAlign on seeded strings -> edlib.npz"""
    rng = np.random.default_rng(20260926)

    def mutate(s, rate):
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

    q_l, t_l, mode_l, out_l, ops_l = [], [], [], [], []
    for it in range(240):
        kind = it % 6
        qn = int(rng.integers(1, 200))
        if it % 7 == 0:
            qn = 64 * int(rng.integers(1, 4))
        if kind < 2:
            t = random_seq(rng, int(rng.integers(1, 320)))
            a = int(rng.integers(0, len(t)))
            q = mutate(t[a:a + qn], float(rng.choice([0.0, 0.02, 0.1, 0.3]))) or b"A"
        elif kind == 2:
            q, t = random_seq(rng, qn), random_seq(rng, int(rng.integers(1, 320)))
        elif kind == 3:
            q, t = random_seq(rng, qn, b"AC"), random_seq(rng, int(rng.integers(1, 100)), b"GT")
        elif kind == 4:
            q, t = random_seq(rng, qn, b"A"), random_seq(rng, int(rng.integers(1, 100)), b"AC")
        else:
            t = random_seq(rng, int(rng.integers(1, 320)))
            q = mutate(t, 0.05) or b"C"
        for mode in (0, 1, 2):
            ed, nl, el, sl, ops = ref.edlib_align(q, t, mode, 2)
            q_l.append(q); t_l.append(t); mode_l.append(mode); out_l.append((ed, nl, el, sl)); ops_l.append(ops)
    d = dict(q=np.array(q_l, dtype=object), t=np.array(t_l, dtype=object), mode=np.array(mode_l, dtype=np.int32),
             out=np.array(out_l, dtype=np.int32), ops=np.array(ops_l, dtype=object))
    # splitAlign: reference window vs consensus with a planted insertion
    c_l, r_l, rc_l, r0_l, r1_l = [], [], [], [], []
    for it in range(60):
        n = int(rng.integers(60, 260))
        refw = random_seq(rng, n, b"ACGT" if it % 5 else b"ACGTN")
        cut = int(rng.integers(n // 3, 2 * n // 3))
        ins = random_seq(rng, int(rng.integers(0, 120))) if it % 6 else refw[max(0, cut - 40):cut]
        a = int(rng.integers(0, n // 4))
        b = int(rng.integers(3 * n // 4, n))
        cons = mutate(refw[a:cut] + ins + refw[cut:b], 0.01) or b"A"
        rc, r0, r1, _ = ref.split_align(cons, refw)
        c_l.append(cons); r_l.append(refw); rc_l.append(rc); r0_l.append(r0); r1_l.append(r1)
    d.update(sa_cons=np.array(c_l, dtype=object), sa_ref=np.array(r_l, dtype=object), sa_rc=np.array(rc_l, dtype=np.int32),
             sa_row0=np.array(r0_l, dtype=object), sa_row1=np.array(r1_l, dtype=object))
    np.savez_compressed(os.path.join(HERE, "edlib.npz"), **d)
    print("edlib ok: %d alignments, %d splitAlign (%d true)" % (len(q_l), len(c_l), int(np.sum(np.array(rc_l) == 1))))


def long_read_vectors(ref):
    """edlib in its Hirschberg regime / with the extended-IUPAC equalities, and msaEdlib -> longread.npz"""
    rng = np.random.default_rng(20260927)

    def ont(s, rate):
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

    q_l, t_l, mode_l, out_l, ops_l = [], [], [], [], []
    for it in range(10):
        tn = int(rng.integers(1500, 3600))
        t = random_seq(rng, tn, b"ACGT" if it % 2 else b"ACGTMRWBSYDKEF-")
        q = ont(bytes(c for c in t if c in b"ACGT"), float(rng.choice([0.02, 0.06, 0.12])))
        for mode in ((0, 16) if it % 2 == 0 else (0, 2, 1)):
            ed, nl, el, sl, ops = ref.edlib_align(q, t, mode, 2)
            q_l.append(q); t_l.append(t); mode_l.append(mode); out_l.append((ed, nl, el, sl)); ops_l.append(ops)
    d = dict(q=np.array(q_l, dtype=object), t=np.array(t_l, dtype=object), mode=np.array(mode_l, dtype=np.int32),
             out=np.array(out_l, dtype=np.int32), ops=np.array(ops_l, dtype=object))
    ref.params = abi.params_lr()
    sets, rows_l, cs_l = [], [], []
    for it in range(6):
        L = int(rng.integers(400, 2300))
        base = random_seq(rng, L + 200)
        n = int(rng.integers(3, 13))
        reads = [ont(base[int(rng.integers(0, 100)):L + 100 + int(rng.integers(0, 100))], 0.06 if it % 2 else 0.02) for _ in range(n)]
        rows, cs = ref.msa_edlib(reads)
        sets.append(np.array(reads, dtype=object)); rows_l.append(rows); cs_l.append(cs)
    # msaWfa (src/assemble.h:547-726): insertion haplotypes, with and without reference anchors
    wsets, wpre, wsuf, wrows, wcs = [], [], [], [], []
    for it in range(5):
        F = int(rng.integers(300, 1200))
        left, right = random_seq(rng, F + 600), random_seq(rng, F + 600)
        hap = left + random_seq(rng, int(rng.integers(100, 700))) + right
        n = int(rng.integers(3, 10))
        reads = [ont(hap[600 - int(rng.integers(0, min(F, 500))):len(hap) - 600 + int(rng.integers(0, min(F, 500)))],
                     0.06 if it % 2 else 0.02) for _ in range(n)]
        pre, suf = (left[-300:], right[:300]) if it % 2 == 0 else (b"", b"")
        rows, cs = ref.msa_wfa(reads, pre, suf)
        wsets.append(np.array(reads, dtype=object)); wpre.append(pre); wsuf.append(suf); wrows.append(rows); wcs.append(cs)
    ref.params = abi.params_sr()
    d.update(msa_sets=np.array(sets, dtype=object), msa_rows=np.array(rows_l, dtype=np.int32), msa_cs=np.array(cs_l, dtype=object),
             wfa_sets=np.array(wsets, dtype=object), wfa_pre=np.array(wpre, dtype=object), wfa_suf=np.array(wsuf, dtype=object),
             wfa_rows=np.array(wrows, dtype=np.int32), wfa_cs=np.array(wcs, dtype=object))
    np.savez_compressed(os.path.join(HERE, "longread.npz"), **d)
    print("longread ok: %d alignments, %d msaEdlib sets, %d msaWfa sets" % (len(q_l), len(sets), len(wsets)))


# split-read genotyping classifier (src/coverage.h:412-434): (label, make_align_jobs kwargs, flank_quality)
ALIGN_JOBS = [("plain", dict(n_sv=40, reads_per_bp=24, seed=11), 0.95),
              ("weird", dict(n_sv=40, reads_per_bp=16, seed=12, weird=True), 0.95),
              ("lowq", dict(n_sv=24, reads_per_bp=16, seed=13, weird=True), 0.4)]


def align_job_vectors(ref):
    d = {}
    for label, kw, fq in ALIGN_JOBS:
        jobs, blob = synth.make_align_jobs(**kw)
        p = abi.params_sr()
        p.flank_quality = fq
        d[label + "_results"] = ref.classify_reads(jobs, blob, params=p)
        d[label + "_jobs"] = jobs       # inputs are stored too, so the fixture does not depend on the generator
        d[label + "_blob"] = blob
    np.savez_compressed(os.path.join(HERE, "align_jobs.npz"), **d)
    print("align_jobs ok:", {k: int(v.shape[0]) for k, v in d.items() if k.endswith("_results")})


def nw_job_vectors(ref):
    """long-read genotyping pairs (src/genotype.h:21-30,276,284): the reference's _editDistanceNW per pair"""
    d = {}
    for label, kw in (("plain", dict(n_reads=48, seed=31)), ("weird", dict(n_reads=45, seed=32, weird=True))):
        jobs, blob = synth.make_nw_jobs(**kw)
        d[label + "_dist"] = ref.edit_distance_nw_batch(jobs, blob)
        d[label + "_jobs"] = jobs
        d[label + "_blob"] = blob
    np.savez_compressed(os.path.join(HERE, "nw_jobs.npz"), **d)
    print("nw_jobs ok:", {k: int(v.shape[0]) for k, v in d.items() if k.endswith("_dist")})


PROBE_BATCHES = [("c2", dict(mode="c2", seed=41), 96), ("mixed", dict(mode="mixed", seed=42), 96), ("ins", dict(mode="ins", seed=43), 96)]


def probe_vectors(ref):
    """per-SV body of _generateProbes (src/coverage.h:196-258) from the reference's own functions"""
    d = {}
    for label, kw, n in PROBE_BATCHES:
        rec, blob = ref.generate_probes(synth.make_batch(n, **kw))
        d[label + "_rec"] = rec
        d[label + "_blob"] = blob
        d[label + "_n"] = n
        d[label + "_kwargs"] = repr(kw)
    np.savez_compressed(os.path.join(HERE, "probes.npz"), **d)
    print("probes ok:", {l: int(d[l + "_rec"]["ok"].sum()) for l, _, _ in PROBE_BATCHES})


def main():
    pyoracle.build()
    ref = pyoracle.Oracle("reference")
    only = sys.argv[1:]  # e.g. "u_ins edlib": regenerate just these (zip timestamps churn otherwise)
    if not only or "edlib" in only:
        edlib_vectors(ref)
    if not only or "longread" in only:
        long_read_vectors(ref)
    if not only or "align_jobs" in only:
        align_job_vectors(ref)
    if not only or "nw_jobs" in only:
        nw_job_vectors(ref)
    if not only or "probes" in only:
        probe_vectors(ref)
    # --- batches ---------------------------------------------------------------
    for name, (n, kw) in BATCHES.items():
        if only and name not in only:
            continue
        b = synth.make_batch(n, **kw)
        lr = name in LR_BATCHES
        res, blob = ref.refine_batch(b, want_alignment=True, params=abi.params_lr(realign=True) if lr else None)
        np.savez_compressed(os.path.join(HERE, "batch_%s.npz" % name), n=n, kwargs=repr(kw), results=res, blob=blob,
                            lr=int(lr))
        print(name, "ok=%d/%d" % (int(res["ok"].sum()), n))
    if only and "primitives" not in only:
        return
    # --- primitives ------------------------------------------------------------
    rng = np.random.default_rng(20260925)
    prim = {}
    # lcs
    a = [random_seq(rng, int(rng.integers(1, 200))) for _ in range(40)]
    b = [random_seq(rng, int(rng.integers(1, 200))) for _ in range(40)]
    for i in range(0, 40, 4):  # related pairs
        x = bytearray(a[i])
        for k in rng.integers(0, len(x), max(1, len(x) // 20)):
            x[k] = rng.choice(list(b"ACGT"))
        b[i] = bytes(x)
    prim["lcs_a"], prim["lcs_b"] = np.array(a, dtype=object), np.array(b, dtype=object)
    prim["lcs_out"] = np.array([ref.lcs(x, y) for x, y in zip(a, b)], dtype=np.int32)
    # reverseComplement incl. lower case / IUPAC quirk (util.h:549-563)
    rc_in = [random_seq(rng, int(rng.integers(1, 60)), b"ACGTNacgtnRYKM-") for _ in range(30)]
    prim["rc_in"] = np.array(rc_in, dtype=object)
    prim["rc_out"] = np.array([ref.reverse_complement(x) for x in rc_in], dtype=object)
    # longestHomology
    ha, hb = [], []
    for _ in range(60):
        L = int(rng.integers(0, 40))
        x = random_seq(rng, L)
        y = bytearray(x + random_seq(rng, int(rng.integers(0, 10))))
        for k in range(len(y)):
            if rng.random() < 0.15:
                y[k] = rng.choice(list(b"ACGT"))
        if rng.random() < 0.3 and len(y) > 2:
            del y[int(rng.integers(0, len(y)))]
        ha.append(x)
        hb.append(bytes(y))
    prim["hom_a"], prim["hom_b"] = np.array(ha, dtype=object), np.array(hb, dtype=object)
    prim["hom_out"] = np.array([ref.longest_homology(x, y) for x, y in zip(ha, hb)], dtype=np.int32)
    # longNeedle on raw strings
    ln_s1, ln_s2, ln_found, ln_r0, ln_r1 = [], [], [], [], []
    for it in range(48):
        n = int(rng.integers(40, 700))
        m = int(rng.integers(10, 200))
        alpha = b"ACGT" if it % 6 else b"ACGTNRacgt"
        s2 = random_seq(rng, n, alpha)
        if it % 5 == 4:
            s1 = random_seq(rng, m, alpha)
        else:
            p = int(rng.integers(0, max(1, n // 2 - m // 2)))
            q = int(rng.integers(n // 2, max(n // 2 + 1, n - m // 2)))
            s1 = bytearray(s2[p:p + m // 2] + s2[q:q + (m - m // 2)])
            for k in rng.integers(0, max(1, len(s1)), max(1, len(s1) // 40)):
                if len(s1):
                    s1[k] = rng.choice(list(b"ACGT"))
            s1 = bytes(s1)
        f, r0, r1, _ = ref.long_needle(s1, s2)
        ln_s1.append(s1); ln_s2.append(s2); ln_found.append(int(f)); ln_r0.append(r0); ln_r1.append(r1)
    prim["ln_s1"], prim["ln_s2"] = np.array(ln_s1, dtype=object), np.array(ln_s2, dtype=object)
    prim["ln_found"] = np.array(ln_found, dtype=np.int32)
    prim["ln_r0"], prim["ln_r1"] = np.array(ln_r0, dtype=object), np.array(ln_r1, dtype=object)
    # msa / guide tree / gotoh / consensus on read sets
    sets, msa_rows, msa_cs, roots, ds, ps = [], [], [], [], [], []
    for it in range(16):
        nr = int(rng.integers(2, 13))
        base = random_seq(rng, 260)
        reads = []
        while len(reads) < nr:
            o = int(rng.integers(0, 110))
            L = int(rng.integers(90, 151))
            x = bytearray(base[o:o + L])
            for k in range(len(x)):
                if rng.random() < 0.01:
                    x[k] = rng.choice(list(b"ACGT"))
            if it % 4 == 3 and rng.random() < 0.5 and len(x) > 20:  # an indel
                del x[int(rng.integers(5, len(x) - 5))]
            if bytes(x) not in reads:
                reads.append(bytes(x))
        rows, cs = ref.msa(reads)
        root, d, p = ref.guide_tree(reads)
        sets.append(np.array(reads, dtype=object)); msa_rows.append(rows); msa_cs.append(cs)
        roots.append(root); ds.append(d); ps.append(p)
    prim["msa_sets"] = np.array(sets, dtype=object)
    prim["msa_rows"] = np.array(msa_rows, dtype=np.int32)
    prim["msa_cs"] = np.array(msa_cs, dtype=object)
    prim["tree_root"] = np.array(roots, dtype=np.int32)
    prim["tree_d"] = np.array(ds, dtype=object)
    prim["tree_p"] = np.array(ps, dtype=object)
    # gotoh: leaf x leaf and profile x profile
    g_a1, g_a2, g_score, g_rows = [], [], [], []
    for it in range(12):
        base = random_seq(rng, 200)
        def var(o, L):
            x = bytearray(base[o:o + L])
            for k in range(len(x)):
                if rng.random() < 0.02:
                    x[k] = rng.choice(list(b"ACGT"))
            return bytes(x)
        x1, x2, x3, x4 = var(0, 120), var(20, 130), var(40, 120), var(10, 150)
        if it % 2 == 0:
            a1, a2 = [x1], [x2]
        else:
            _, a1 = ref.gotoh([x1], [x2])
            _, a2 = ref.gotoh([x3], [x4])
        sc, rows = ref.gotoh(a1, a2)
        g_a1.append(np.array(a1, dtype=object)); g_a2.append(np.array(a2, dtype=object))
        g_score.append(sc); g_rows.append(np.array(rows, dtype=object))
    prim["gotoh_a1"] = np.array(g_a1, dtype=object)
    prim["gotoh_a2"] = np.array(g_a2, dtype=object)
    prim["gotoh_score"] = np.array(g_score, dtype=np.int32)
    prim["gotoh_rows"] = np.array(g_rows, dtype=object)
    prim["gotoh_cons"] = np.array([ref.consensus(list(r)) for r in g_rows], dtype=object)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **prim)
    print("primitives ok")


if __name__ == "__main__":
    main()
