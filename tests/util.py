"""Shared comparison helpers for the parity tests."""
import numpy as np

import pyoracle

# fields that every implementation must agree on bit-for-bit
CORE = ["svid", "ok", "sv_start", "sv_end", "ci_wiggle", "ins_len", "cons_bp", "hom_len", "sr_support",
        "sr_align_quality", "c_start", "c_end", "r_start", "r_end", "hom_left", "hom_right", "cons_len",
        "ref_len", "allele_len", "aln_len", "status"]
# internals the reference API does not expose (port vs HIP only)
INTERNAL = ["score_unsplit", "matches", "mismatches"]
INTERNAL_FOUND = ["score_best", "cons_left", "ref_left", "ref_right"]
SCORE_UNKNOWN = -(1 << 30)   # dh::SP_UNKNOWN


def compare(res_a, blob_a, res_b, blob_b, fields=CORE, blobs=("cons", "allele", "aln"), label=""):
    bad = []
    assert res_a.shape == res_b.shape
    for f in fields:
        x, y = res_a[f], res_b[f]
        if x.dtype.kind == "f":
            neq = ~((x == y) | (np.isnan(x) & np.isnan(y)))
        else:
            neq = x != y
        if f == "score_unsplit":
            # the sparse longNeedle (sparse_needle.hpp) reports mat[m][n] only when it lies within its deficit budget:
            # below it the value cannot change the result (bestScore > mat[m][n] is already certain)
            neq &= (x != SCORE_UNKNOWN) & (y != SCORE_UNKNOWN)
        for i in np.nonzero(neq)[0][:5]:
            bad.append("%s junction %d field %s: %r vs %r" % (label, i, f, x[i], y[i]))
    for i in range(res_a.shape[0]):
        for w in blobs:
            if pyoracle.blob_field(res_a[i], blob_a, w) != pyoracle.blob_field(res_b[i], blob_b, w):
                bad.append("%s junction %d blob %s differs" % (label, i, w))
                if len(bad) > 20:
                    break
    assert not bad, "\n".join(bad[:20])


def compare_probes(a, ab, b, bb, label=""):
    """dellyhip_probes records + probe bytes (offsets are layout, not content)"""
    import numpy as np
    assert a.shape == b.shape, label
    for f in a.dtype.names:
        if "_off" in f or f == "reserved":
            continue
        bad = np.nonzero(a[f] != b[f])[0]
        assert bad.size == 0, (label, f, bad[:5], a[f][bad[:5]], b[f][bad[:5]])
    for k in np.nonzero(a["ok"])[0]:
        for w in ("cons", "ref"):
            for bp in "01":
                n = int(a[w + "_len" + bp][k])
                oa, ob = int(a[w + "_off" + bp][k]), int(b[w + "_off" + bp][k])
                assert bytes(ab[oa:oa + n]) == bytes(bb[ob:ob + n]), (label, int(k), w, bp)


def compare_compact(res, blob, ref_res, ref_blob, batch, label=""):
    """records of a compact-payload run (dellyhip_params.reserved bit 2: allele_len = -(length), no "REF,ALT" bytes) against the
    reference: every CORE field but the allele bookkeeping, the consensus bytes, and the alleles RE-CUT on the host from the record
    (dellyhip_recut_alleles) against the reference's own"""
    from delly_amd import abi, refine
    import numpy as np
    compare(res, blob, ref_res, ref_blob, fields=[f for f in CORE if f != "allele_len"], blobs=("cons",), label=label)
    assert (np.abs(res["allele_len"]) == ref_res["allele_len"]).all(), label
    recut = refine.recut_alleles(abi.params_sr(compact_alleles=True), batch.junctions, np.ascontiguousarray(res), np.ascontiguousarray(blob), batch.chroms)
    for i in range(res.shape[0]):
        got = recut[i] if res["allele_len"][i] < 0 else pyoracle.blob_field(res[i], blob, "allele")
        assert got == pyoracle.blob_field(ref_res[i], ref_blob, "allele"), "%s junction %d: re-cut alleles differ" % (label, i)
