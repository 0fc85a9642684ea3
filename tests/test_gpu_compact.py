"""-m gpu: the compact result payload (dellyhip_params.reserved bit 2, DELLYHIP_COMPACT_ALLELES).  Where the exact alleles of
src/split.h:606-624 are plain substrings of the reference window and the consensus, the kernels return allele_len = -(length)
and no bytes; dellyhip_recut_alleles rebuilds "REF,ALT" on the host from c_start / c_end / r_start / r_end.  The re-cut alleles
must equal the reference's (oracle/_ref) byte for byte, every other field must be what the full payload gives, and the blob
must shrink by exactly the omitted bytes."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare
import pyoracle

pytestmark = pytest.mark.gpu
THREADS = os.cpu_count() or 1


def _alleles(res, blob, recut):
    out = []
    for i in range(res.shape[0]):
        out.append(recut[i] if res["allele_len"][i] < 0 else pyoracle.blob_field(res[i], blob, "allele"))
    return out


@pytest.mark.parametrize("mode,n,kw", [("c2", 3000, {}), ("c2", 1500, dict(sub_rate=0.02)), ("mixed", 900, {}), ("ins", 400, {}),
                                        ("c2", 600, dict(genome="lowcx"))])
def test_recut_alleles_equal_the_reference(gpu_ctx, reference, mode, n, kw):
    kw = dict(kw)
    if kw.get("genome"):
        kw["real"] = synth.load_real_chromosome()
    b = synth.make_batch(n, mode=mode, seed=31, **kw)
    params = abi.params_sr(compact_alleles=True)
    ctx = refine.Context(params=params)
    try:
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=False)
        gpu_ctx.set_chromosomes(b.chroms)
        fr, fb = gpu_ctx.refine(b, want_alignment=False)          # the full payload, same batch
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        # records: everything but the allele bytes' bookkeeping is as the reference has it
        compare(gr, gb, rr, rb, fields=[f for f in CORE if f != "allele_len"], blobs=("cons",), label="compact " + mode)
        compact = gr["allele_len"] < 0
        assert (np.abs(gr["allele_len"]) == rr["allele_len"]).all()
        if mode == "c2":
            assert compact.sum() > 0.9 * (rr["allele_len"] > 0).sum()       # the sparse kernel's junctions: all compact
        recut = refine.recut_alleles(params, b.junctions, gr, gb, b.chroms)
        got = _alleles(gr, gb, recut)
        for i in range(n):
            assert got[i] == pyoracle.blob_field(rr[i], rb, "allele"), (mode, i, got[i][:40])
        # the blob lost exactly the omitted bytes
        assert fb.nbytes - gb.nbytes == int(-gr["allele_len"][compact].sum())
        assert (gr["allele_off"][compact] == 0).all()
    finally:
        ctx.close()


def test_recut_argument_errors(gpu_ctx):
    b = synth.make_batch(200, mode="c2", seed=2)
    params = abi.params_sr(compact_alleles=True)
    ctx = refine.Context(params=params)
    try:
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=False)
        k = int(np.nonzero(gr["allele_len"] < 0)[0][0])
        bad = gr.copy()
        bad["r_end"][k] += 1                                   # a record that does not add up is refused, not guessed at
        with pytest.raises(refine.DellyHipError):
            refine.recut_alleles(params, b.junctions[k:k + 1], bad[k:k + 1], gb, b.chroms)
        short = [c[:100] for c in b.chroms]                    # the caller's chromosome is shorter than the window
        with pytest.raises(refine.DellyHipError):
            refine.recut_alleles(params, b.junctions[k:k + 1], gr[k:k + 1], gb, short)
    finally:
        ctx.close()


def test_stream_with_the_compact_payload(gpu_ctx, reference):
    """dellyhip_stream (host buffers in, host buffers out) inherits the flag from its context: what bench.py's host_inclusive leg times"""
    b = synth.make_batch(4000, mode="c2", seed=17)
    params = abi.params_sr(compact_alleles=True)
    ctx = refine.Context(params=params)
    try:
        ctx.set_chromosomes(b.chroms)
        st = refine.Stream(ctx, depth=3)
        st.submit(b, tag=5)
        gr, gb, tag = st.collect()
        gr, gb = np.array(gr), np.array(gb)
        st.close()
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        compare(gr, gb, rr, rb, fields=[f for f in CORE if f != "allele_len"], blobs=("cons",), label="compact stream")
        recut = refine.recut_alleles(params, b.junctions, gr, gb, b.chroms)
        for i in range(b.n):
            want = pyoracle.blob_field(rr[i], rb, "allele")
            assert (recut[i] if gr["allele_len"][i] < 0 else pyoracle.blob_field(gr[i], gb, "allele")) == want, i
        assert gb.nbytes < 0.45 * rb.nbytes or rb.nbytes == 0
    finally:
        ctx.close()


def test_stream_reads_pinned_sequence_bytes_in_place(gpu_ctx, reference):
    """dellyhip_stream_zero_copy: a pinned seq_blob is not staged; same records.  A pageable blob on the same stream is staged as before."""
    bs = [synth.make_batch(3000, mode="c2", seed=s) for s in (41, 42, 43)]
    import bench
    chroms, batches = bench.one_genome(synth, bs)
    gpu_ctx.set_chromosomes(chroms)
    st = refine.Stream(gpu_ctx, depth=3)
    st.zero_copy(True)
    keep = []
    for k, b in enumerate(batches):
        blob = np.ascontiguousarray(b.seq_blob, dtype=np.uint8).copy()
        if k != 1:                                  # batch 1 stays pageable
            gpu_ctx.host_register(blob.ctypes.data, blob.nbytes)
        keep.append(blob)
        b2 = synth.Batch(b.chroms, b.junctions, blob, b.seq_off, b.with_msa, b.truth)
        st.submit(b2, tag=k)
    for k, b in enumerate(batches):
        gr, gb, tag = st.collect()
        assert tag == k
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        compare(gr, gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label="zero-copy stream batch %d" % k)
    st.close()
    for k, blob in enumerate(keep):
        if k != 1:
            gpu_ctx.host_unregister(blob.ctypes.data)
