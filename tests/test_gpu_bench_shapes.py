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
             "sr_stage_mixed_all_svt": None, "lr_c4_align_consensus": None, "lr_c4_msaedlib_n15": None, "lr_ins_msawfa_n15": None,
             # 10 kb x 20.7 kb: the reference needs 3.3 GB and ~3 s per junction -- 8 threads, ~25 s for all 64 benched junctions
             # (round 5 compared 16 of them)
             "lr_stress_10kb_x_20kb": None}
TILED = [x[0] for x in bench.SIDE_PLAN if int(x[3].get("_tiles", 1)) > 1]


def _check(ctx, ref, b, params, label, n_cmp=None, ref_threads=None):
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=False)
    sub = b if n_cmp is None or n_cmp >= b.n else bench._subbatch(b, n_cmp)
    lr = params is not None and (params.reserved & 1)
    rr, rb = ref.refine_batch(sub, want_alignment=False, n_threads=min(THREADS, ref_threads or 64) if lr else THREADS, params=params)
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


def test_one_context_on_alternating_streams_serialises_its_runs(gpu_ctx):
    """Runs of ONE context share its scratch area and work counters, so a run on another stream than the previous one has to wait
    for it: the context records its serialisation event on the previous run's stream only then (not after every run -- an event
    record is 5 - 7 us of stream time).  Three batches, short and long ones mixed, alternate between the two compute streams without
    host waits; every batch must come out as it does alone."""
    # (2 % substitutions: every junction needs several deficit levels, whose tables pass through the wavefront's scratch area.  The
    #  test exercises the record-then-wait path; it does not PROVE the serialisation -- a build that never waits passed it too on the
    #  one box it was tried on: the second launch's workgroups only start as the first one's leave, and rarely next to their namesakes)
    raw = [synth.make_batch(n, mode="c2", first=k * 10000, seed=40 + k, sub_rate=0.02) for k, n in enumerate((6000, 300, 9000))]
    chroms, batches = bench.one_genome(synth, raw)
    gpu_ctx.set_chromosomes(chroms)
    alone = [gpu_ctx.refine(b) for b in batches]
    streams = gpu_ctx.compute_streams()
    rbs = [gpu_ctx.upload(b) for b in batches]
    order = [0, 1, 2, 1, 0, 2, 2, 1, 0]
    for i, k in enumerate(order):
        rbs[k].run(streams[i % 2])
    for k, rb in enumerate(rbs):
        rb.sync()
        r, bl = rb.fetch()
        assert all((r[f] == alone[k][0][f]).all() for f in r.dtype.names), "batch %d" % k
        assert bl.tobytes() == alone[k][1].tobytes()
        rb.free()


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


@pytest.mark.parametrize("name", [x[0] for x in bench.SIDE_PLAN if x[0] in COMPARE_N])   # (every row that is not a tile of another)
def test_side_measurement_batches_vs_reference(reference, name):
    _, n, _, kw = [x for x in bench.SIDE_PLAN if x[0] == name][0]
    lr = kw["mode"].startswith("lr")
    params = abi.params_lr(realign=True) if lr else abi.params_sr()
    ctx = refine.Context(params=params)
    try:
        b = bench.side_batch(synth, n, kw)
        gr = _check(ctx, reference, b, params, name, COMPARE_N[name], ref_threads=kw.get("_cpu_threads"))
        assert int((gr["status"] != 0).sum()) == 0
        assert int(gr["ok"].sum()) > 0.75 * n
        if name == "sr_stage_mixed_all_svt":   # BASELINE configs[2]: every SV type, 2 .. 20 reads
            assert sorted(set(b.junctions["svt"].tolist())) == list(range(9))
            assert b.junctions["n_seq"].min() == 2 and b.junctions["n_seq"].max() == 20 and len(b.chroms) == 2
        if name == "lr_stress_10kb_x_20kb":
            assert gr["cons_len"].min() > 9000 and gr["ref_len"].min() > 20000
    finally:
        ctx.close()


@pytest.mark.parametrize("name", TILED)
def test_chip_filling_rows_are_tiles_of_a_compared_batch(name):
    """the chip-filling long-read rows are `_tiles` copies of their sibling row's batch (compared in full with the reference above)
    side by side on a longer genome: every tile must come out like the first, shifted by the tile's offset"""
    _, n, _, kw = [x for x in bench.SIDE_PLAN if x[0] == name][0]
    tiles = int(kw["_tiles"])
    params = abi.params_lr(realign=True)
    ctx = refine.Context(params=params)
    try:
        b = bench.side_batch(synth, n, kw)
        base = [x for x in bench.SIDE_PLAN if x[3].get("mode") == kw["mode"] and x[3].get("n_reads") == kw.get("n_reads")
                and x[3].get("sub_rate") == kw.get("sub_rate") and "_tiles" not in x[3] and "_big" not in x[3]]
        assert base and base[0][1] == n // tiles, "a tiled row needs its sibling row of n / tiles junctions"
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=False)
        n0 = n // tiles
        csize = b.chroms[0].size // tiles
        first = gr[:n0].copy()
        for k in range(1, tiles):
            want = first.copy()
            want["svid"] += k * n0
            moved = want["ok"] != 0
            want["sv_start"][moved] += k * csize
            want["sv_end"][moved] += k * csize
            got = gr[k * n0:(k + 1) * n0]
            for f in CORE:
                if f in ("sv_start", "sv_end"):   # (an unrefined junction keeps the caller's coordinates, which are shifted too)
                    assert (got[f][moved] == want[f][moved]).all(), (name, k, f)
                else:
                    x, y = got[f], want[f]
                    same = (x == y) | ((x != x) & (y != y)) if x.dtype.kind == "f" else (x == y)
                    assert same.all(), (name, k, f)
            import pyoracle
            for i in range(0, n0, max(1, n0 // 64)):
                for w in ("cons", "allele"):
                    assert pyoracle.blob_field(got[i], gb, w) == pyoracle.blob_field(gr[i], gb, w), (name, k, i, w)
        assert int((gr["status"] != 0).sum()) == 0
    finally:
        ctx.close()
