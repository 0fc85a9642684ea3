"""-m gpu: dellyhip_stream, the pipelined host-buffer path (pinned staging, recycled device buffers, one H2D copy per batch,
device-side compaction, routing of the sparse kernel's leftovers deferred to collect()).  Every batch of a stream must
come back byte-identical to the classic upload / run / fetch path and to the oracle, whatever the order, the size of the
neighbouring batches, or the path a junction takes (sparse kernel, dense leftovers, insertions, MSA, long reads)."""
import numpy as np
import pytest

import fuzz
from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu


def _drive(ctx, batches, depth, with_msa, want):
    """keeps `depth` batches in flight; collect() copies and releases"""
    st = refine.Stream(ctx, depth=depth, with_msa=with_msa, want_alignment=want)
    got, nxt = {}, 0
    for k in range(len(batches)):
        while nxt < len(batches) and nxt - k < depth:
            st.submit(batches[nxt], tag=nxt)
            nxt += 1
        res, blob, tag = st.collect()
        assert tag == k
        got[k] = (res, blob)
    assert st.pending() == 0
    st.close()
    return got


def _noisy(batch, rate, seed):
    rng = np.random.default_rng(seed)
    blob = batch.seq_blob.copy()
    hit = rng.random(blob.size) < rate
    blob[hit] = synth.ACGT[rng.integers(0, 4, int(hit.sum()))]
    return synth.Batch(batch.chroms, batch.junctions, blob, batch.seq_off, batch.with_msa, batch.truth)


def _one_genome(batches):
    """batches made independently -> one chromosome table (concatenated), coordinates shifted"""
    chroms = [np.concatenate([b.chroms[c] for b in batches]) for c in range(len(batches[0].chroms))]
    out, base = [], [0] * len(chroms)
    for b in batches:
        j = b.junctions.copy()
        j["sv_start"] += base[0]
        j["sv_end"] += np.where(j["chr2"] == 0, base[0], base[-1])
        out.append(synth.Batch(chroms, j, b.seq_blob, b.seq_off, b.with_msa, b.truth))
        base = [x + c.size for x, c in zip(base, b.chroms)]
    return chroms, out


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_stream_unit_u_matches_oracle_and_classic(port, depth):
    sizes = [700, 40, 1300, 0, 1, 900, 2500, 64]
    raw = [synth.make_batch(n, mode="c2", seed=50 + i) for i, n in enumerate(sizes)]
    raw[2] = _noisy(raw[2], 0.2, 3)      # deficits beyond the sparse kernel's 32 levels: dense leftovers at collect()
    raw[5] = _noisy(raw[5], 0.03, 4)
    chroms, batches = _one_genome([b if b.n else synth.make_batch(1, mode="c2", seed=99) for b in raw])
    batches = [b if raw[i].n else synth.subset(b, []) for i, b in enumerate(batches)]
    ctx = refine.Context()
    ctx.set_chromosomes(chroms)
    got = _drive(ctx, batches, depth, 0, False)
    for k, b in enumerate(batches):
        if b.n == 0:
            assert got[k][0].shape[0] == 0
            continue
        pr, pb = port.refine_batch(b)
        compare(got[k][0], got[k][1], pr, pb, fields=[f for f in CORE + INTERNAL if f != "aln_len"], blobs=("cons", "allele"),
                label="stream U batch %d" % k)
        assert (got[k][0]["reserved"] == 0).all()
    ctx.close()


@pytest.mark.parametrize("mode,want", [("mixed", False), ("mixed", True), ("ins", True)])
def test_stream_mixed_types_and_insertions(port, mode, want):
    raw = [fuzz.perturbed(300, 20 + i, mode) for i in range(4)]
    chroms, batches = _one_genome(raw)
    ctx = refine.Context()
    ctx.set_chromosomes(chroms)
    got = _drive(ctx, batches, 3, 0, want)
    for k, b in enumerate(batches):
        pr, pb = port.refine_batch(b)
        compare(got[k][0], got[k][1], pr, pb, fields=[f for f in CORE + INTERNAL if want or f != "aln_len"],
                blobs=("cons", "allele", "aln") if want else ("cons", "allele"), label="stream %s batch %d" % (mode, k))
    ctx.close()


def test_stream_shapes_beyond_the_sparse_kernel(port):
    """consensus of 255 .. 319 bp and windows beyond 1280 bp never enter the sparse list: dense bins at submit time"""
    raw = [synth.make_batch(200, mode="c2", seed=70 + i, cons_flank=140 + 5 * i, del_len=600 + 150 * i) for i in range(3)]
    chroms, batches = _one_genome(raw)
    ctx = refine.Context()
    ctx.set_chromosomes(chroms)
    got = _drive(ctx, batches, 2, 0, False)
    for k, b in enumerate(batches):
        pr, pb = port.refine_batch(b)
        assert int(pr["ok"].sum()) > 150
        compare(got[k][0], got[k][1], pr, pb, fields=[f for f in CORE + INTERNAL if f != "aln_len"], blobs=("cons", "allele"), label="stream big %d" % k)
    ctx.close()


@pytest.mark.parametrize("n_reads,mode", [(5, "c2"), (12, "c2"), (6, "mixed")])
def test_stream_msa_batches(port, n_reads, mode):
    raw = [synth.make_batch(n, mode=mode, n_reads=n_reads, seed=80 + i) for i, n in enumerate([150, 20, 260])]
    raw[2] = _noisy(raw[2], 0.04, 5)     # some consensus sequences the sparse kernel gives up on -> routed at collect()
    chroms, batches = _one_genome(raw)
    ctx = refine.Context()
    ctx.set_chromosomes(chroms)
    got = _drive(ctx, batches, 3, 1, False)
    for k, b in enumerate(batches):
        pr, pb = port.refine_batch(b)
        compare(got[k][0], got[k][1], pr, pb, fields=[f for f in CORE if f != "aln_len"], blobs=("cons", "allele"), label="stream msa %d" % k)
    ctx.close()


def test_stream_msa_with_insertions_takes_the_routed_path(port):
    b = synth.make_batch(120, mode="ins", n_reads=6, seed=91)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    got = _drive(ctx, [b, b], 2, 1, True)
    pr, pb = port.refine_batch(b)
    for k in (0, 1):
        compare(got[k][0], got[k][1], pr, pb, fields=CORE, label="stream msa+ins %d" % k)
    ctx.close()


def test_stream_long_read_loop_body(port):
    P = abi.params_lr(realign=True)
    raw = [synth.make_batch(6, mode="lr", n_reads=5, sub_rate=0.06, seed=95 + i) for i in range(2)]
    chroms, batches = _one_genome(raw)
    ctx = refine.Context(params=P)
    ctx.set_chromosomes(chroms)
    got = _drive(ctx, batches, 2, 2, False)
    for k, b in enumerate(batches):
        pr, pb = port.refine_batch(b, params=P)
        compare(got[k][0], got[k][1], pr, pb, fields=[f for f in CORE if f != "aln_len"], blobs=("cons", "allele"), label="stream lr %d" % k)
    ctx.close()


def test_stream_slot_discipline_and_shared_chromosomes():
    b = synth.make_batch(50, mode="c2", seed=7)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    st = refine.Stream(ctx, depth=2)
    st.submit(b)
    st.submit(b)
    with pytest.raises(refine.DellyHipError):
        st.submit(b)                       # both slots in flight
    r0, _, _ = st.collect(copy=False)
    with pytest.raises(refine.DellyHipError):
        st.submit(b)                       # slot 0 is held by the caller until the next collect
    ok0 = int(r0["ok"].sum())
    r1, _, _ = st.collect()
    st.submit(b)                           # slot 0 released by the second collect
    r2, _, _ = st.collect()
    assert ok0 == int(r1["ok"].sum()) == int(r2["ok"].sum()) > 40
    with pytest.raises(refine.DellyHipError):
        st.collect()                       # nothing submitted
    st.close()
    # a context sharing the chromosome table: no upload of its own
    c2 = refine.Context(share_with=ctx)
    r3, _ = c2.refine(b)
    assert int(r3["ok"].sum()) == ok0
    c2.close()
    ctx.close()


def _same(a, b):
    """record arrays equal field by field (numpy does not carry the padding bytes of the record through .copy())"""
    return a.shape == b.shape and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def test_stream_survives_a_rejected_batch_and_an_empty_one():
    good = synth.make_batch(300, mode="c2", seed=21)
    ctx = refine.Context()
    ctx.set_chromosomes(good.chroms)
    want_res, want_blob = ctx.refine(good)
    bad_j = good.junctions.copy()
    bad_j["sv_start"][5] = -3                # malformed record: rejected before any kernel sees the batch
    bad = synth.Batch(good.chroms, bad_j, good.seq_blob, good.seq_off, good.with_msa, good.truth)
    empty = synth.Batch(good.chroms, good.junctions[:0].copy(), good.seq_blob[:0].copy(), good.seq_off[:1].copy(),
                        good.with_msa, None)
    st = refine.Stream(ctx, depth=2)
    st.submit(good, tag=1)
    with pytest.raises(refine.DellyHipError) as e:
        st.submit(bad, tag=2)
    assert e.value.code == abi.E_ARG
    assert st.pending() == 1                 # the rejected batch took no slot
    st.submit(empty, tag=3)
    st_res, st_blob, tag = st.collect()
    assert tag == 1 and _same(st_res, want_res) and st_blob.tobytes() == want_blob.tobytes()
    r, b, tag = st.collect()
    assert tag == 3 and r.size == 0 and b.size == 0
    st.submit(good, tag=4)                   # and the stream goes on
    st_res, st_blob, tag = st.collect()
    assert tag == 4 and _same(st_res, want_res) and st_blob.tobytes() == want_blob.tobytes()
    st.close()
    ctx.close()


def test_stream_holds_its_memory_over_many_batches_of_changing_size():
    import torch
    sizes = [900, 40, 2500, 1, 700, 1800]
    batches = [synth.make_batch(n, mode="c2", seed=30 + i) for i, n in enumerate(sizes)]
    chroms, batches = _one_genome(batches)
    ctx = refine.Context()
    ctx.set_chromosomes(chroms)
    classic = [ctx.refine(b) for b in batches]
    st = refine.Stream(ctx, depth=3)

    def lap():
        got = []
        for k in range(len(batches) + 2):
            if k < len(batches):
                st.submit(batches[k], tag=k)
            if k >= 2:
                got.append(st.collect())
        return got

    for _ in range(2):                       # slots reach their high-water sizes
        lap()
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        got = lap()
    torch.cuda.synchronize()
    after = torch.cuda.mem_get_info()[0]
    assert before - after < (4 << 20), "stream grew by %d bytes over 36 batches" % (before - after)
    for k, (r, bl, tag) in enumerate(got):
        assert tag == k and _same(r, classic[k][0]) and bl.tobytes() == classic[k][1].tobytes()
    st.close()
    ctx.close()


def test_two_host_threads_each_with_a_stream_on_one_genome():
    """the worker-thread model of src/shortpe.h:175-201 on the pipelined path: one resident genome, one dellyhip_stream per
    host thread (contexts from dellyhip_create_shared), submitting and collecting concurrently -- the library's pools of
    parked memory and its per-device HIP streams are shared between them"""
    import threading
    raw = [synth.make_batch(700 + 150 * i, mode="mixed", seed=60 + i) for i in range(6)]
    chroms, batches = _one_genome(raw)
    root = refine.Context()
    root.set_chromosomes(chroms)
    classic = [root.refine(b) for b in batches]
    errors, got = [], {}

    def worker(tid):
        try:
            ctx = refine.Context(share_with=root)
            st = refine.Stream(ctx, depth=3)
            mine = [k for k in range(len(batches)) if k % 2 == tid]
            for lap in range(4):
                nxt = 0
                for i in range(len(mine)):
                    while nxt < len(mine) and nxt - i < 2:
                        st.submit(batches[mine[nxt]], tag=mine[nxt])
                        nxt += 1
                    r, bl, tag = st.collect()
                    got[(tid, lap, int(tag))] = (r, bl)
            st.close()
            ctx.close()
        except Exception as e:   # noqa: BLE001 (reported by the main thread)
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(got) == 2 * 4 * 3
    for (tid, lap, k), (r, bl) in got.items():
        assert _same(r, classic[k][0]) and bl.tobytes() == classic[k][1].tobytes(), (tid, lap, k)
    root.close()
