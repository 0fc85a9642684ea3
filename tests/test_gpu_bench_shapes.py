"""-m gpu: bit-compare of the HIP path with the reference itself (oracle/_ref, the host's threads) on EXACTLY the
batches bench.py times, each in FULL -- the RESIDENT_BATCHES headline batches of 10 000 C2 junctions, every point of
the deficit sweep (bench.SWEEP_PLAN), and the side measurements of bench.SIDE_PLAN (40 000 C2 junctions, U_full N = 20 /
5, insertions, long-read alignConsensus, msaEdlib and msaWfa loop bodies)."""
import os
import sys

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (SIDE_PLAN only; main() is not run)

THREADS = os.cpu_count() or 1
# junctions compared per workload (None = the whole benched batch); long-read legs use at most 64 reference threads (each
# holds four int32 matrices of ~60 MB, src/needle.h:52-103)
COMPARE_N = {"u_c2_40k_junctions": None, "u_full_n20": None, "u_full_n20_10k_junctions": None, "u_full_n5": None, "ins_svt4": None,
             "lr_c4_align_consensus": None, "lr_c4_msaedlib_n15": None, "lr_ins_msawfa_n15": None}


def _check(ctx, ref, b, params, label, n_cmp=None):
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=False)
    sub = b if n_cmp is None or n_cmp >= b.n else bench._subbatch(b, n_cmp)
    lr = params is not None and (params.reserved & 1)
    rr, rb = ref.refine_batch(sub, want_alignment=False, n_threads=min(THREADS, 64) if lr else THREADS, params=params)
    k = sub.n
    compare(gr[:k], gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label=label)
    return gr


@pytest.mark.parametrize("k", range(bench.RESIDENT_BATCHES))
def test_headline_batches_10000_c2_junctions_vs_reference(gpu_ctx, reference, k):
    """the resident batches bench.py's timed steps rotate through (rank 0 of a one-GPU run)"""
    b = synth.make_batch(10000, mode="c2", first=k * 10000)
    gr = _check(gpu_ctx, reference, b, None, "bench headline batch %d (10 000 C2)" % k)
    assert int(gr["ok"].sum()) >= 9890   # bench.py's refined_ok (1 % pure-reference junctions per batch)


def test_headline_arrangement_two_launches_in_flight_gives_the_same_records(gpu_ctx):
    """bench.py's N = 1 timed region: the resident batches alternate between two contexts that share one genome, on the two
    compute streams of dellyhip_compute_streams, launched back to back without waiting -- every batch must come out as it
    does alone (the batches themselves are compared with oracle/_ref above)"""
    raw = [synth.make_batch(10000, mode="c2", first=k * 10000) for k in range(bench.RESIDENT_BATCHES)]
    chroms, batches = bench.one_genome(synth, raw)
    gpu_ctx.set_chromosomes(chroms)
    alone = [gpu_ctx.refine(b) for b in batches]
    other = refine.Context(share_with=gpu_ctx)
    ctxs = [gpu_ctx, other]
    streams = gpu_ctx.compute_streams()
    assert streams[0] and streams[1] and streams[0] != streams[1]
    rbs = [ctxs[k % 2].upload(b) for k, b in enumerate(batches)]
    for lap in range(3):
        for k, rb in enumerate(rbs):
            rb.run(streams[k % 2])
    for k, rb in enumerate(rbs):
        rb.sync()
        r, bl = rb.fetch()
        assert all((r[f] == alone[k][0][f]).all() for f in r.dtype.names), "batch %d" % k
        assert bl.tobytes() == alone[k][1].tobytes()
        rb.free()
    other.close()


@pytest.mark.parametrize("name", [x[0] for x in bench.SWEEP_PLAN])
def test_deficit_sweep_batches_vs_reference(gpu_ctx, reference, name):
    kw = [x[1] for x in bench.SWEEP_PLAN if x[0] == name][0]
    b = bench.sweep_batch(synth, kw)
    _check(gpu_ctx, reference, b, None, "deficit sweep " + name)


def test_host_inclusive_stream_results_vs_reference(gpu_ctx, reference):
    """what bench.py's host_inclusive leg hands back (dellyhip_stream over the resident batches' host copies)"""
    raw = [synth.make_batch(10000, mode="c2", first=k * 10000) for k in range(2)]
    chroms, batches = bench.one_genome(synth, raw)
    gpu_ctx.set_chromosomes(chroms)
    st = refine.Stream(gpu_ctx, depth=3)
    st.submit(batches[0], tag=0)
    st.submit(batches[1], tag=1)
    for k in range(2):
        gr, gb, tag = st.collect()
        assert tag == k
        rr, rb = reference.refine_batch(batches[k], want_alignment=False, n_threads=THREADS)
        compare(gr, gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label="host-inclusive stream batch %d" % k)
    st.close()


@pytest.mark.parametrize("name", [x[0] for x in bench.SIDE_PLAN if x[0] in COMPARE_N])
def test_side_measurement_batches_vs_reference(reference, name):
    _, n, _, kw = [x for x in bench.SIDE_PLAN if x[0] == name][0]
    lr = kw["mode"].startswith("lr")
    params = abi.params_lr(realign=True) if lr else abi.params_sr()
    ctx = refine.Context(params=params)
    try:
        b = synth.make_batch(n, **kw)
        gr = _check(ctx, reference, b, params, name, COMPARE_N[name])
        assert int((gr["status"] != 0).sum()) == 0
        assert int(gr["ok"].sum()) > 0.75 * n
    finally:
        ctx.close()
