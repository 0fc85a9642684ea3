"""-m gpu: host-side robustness of the C-ABI (round-1 advisor findings): device memory is returned by every
entry point, a resident batch gives the same records on every run, malformed junction records are rejected
with DELLYHIP_E_ARG before any kernel sees them."""
import os

import numpy as np
import pytest
import torch

from delly_amd import abi, refine, synth

pytestmark = pytest.mark.gpu


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_single_call_entry_points_return_their_device_memory(gpu_ctx):
    jobs, jblob = synth.make_align_jobs(40, 40, seed=5)
    nj, nblob = synth.make_nw_jobs(64, seed=5)
    b = synth.make_batch(64, mode="c2", seed=3)
    gpu_ctx.set_chromosomes(b.chroms)
    reads = [bytes(x) for x in np.random.default_rng(1).choice(list(b"ACGT"), (4, 300)).astype(np.uint8)]

    def once():
        gpu_ctx.classify_reads(jobs, jblob)
        gpu_ctx.edit_distance_nw_batch(nj, nblob)
        gpu_ctx.generate_probes(b)
        gpu_ctx.edlib_align(reads[0], reads[1], 0, 0)
        gpu_ctx.edlib_align(reads[0][:100], reads[1][:200], 2, 2)
        gpu_ctx.msa_edlib(reads)
        gpu_ctx.msa_wfa(reads)

    for _ in range(3):   # warm the allocator (context scratch, code objects)
        once()
    before = _free_bytes()
    for _ in range(25):
        once()
    after = _free_bytes()
    assert before - after < (4 << 20), "device memory leaked: %d bytes over 25 rounds" % (before - after)


def test_resident_batch_rerun_is_identical_and_fetch_needs_a_run(gpu_ctx):
    # junction 0 exceeds the short-read window limit only through its coordinates: consensus of 300 bp, window > 2048
    b = synth.make_batch(32, mode="mixed", seed=2)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    with pytest.raises(refine.DellyHipError) as e:
        rb.fetch()
    assert e.value.code == abi.E_ARG
    rb.run(); rb.sync()
    r1, b1 = rb.fetch()
    rb.run(); rb.sync()
    r2, b2 = rb.fetch()
    assert r1.tobytes() == r2.tobytes() and b1.tobytes() == b2.tobytes()
    rb.free()


def test_malformed_junction_records_are_rejected(gpu_ctx):
    b = synth.make_batch(4, mode="c2", seed=7)
    gpu_ctx.set_chromosomes(b.chroms)
    clen = b.chroms[0].size
    for field, val in (("sv_start", -5), ("sv_end", -1), ("sv_start", clen + 1), ("sv_end", clen + 10)):
        j = b.junctions.copy()
        j[field][1] = val
        with pytest.raises(refine.DellyHipError) as e:
            gpu_ctx.align_consensus_batch(j, b.seq_blob, b.seq_off)
        assert e.value.code == abi.E_ARG, field
    off = b.seq_off.copy()
    off[2] = off[1] - 1   # not monotonic
    with pytest.raises(refine.DellyHipError) as e:
        gpu_ctx.align_consensus_batch(b.junctions, b.seq_blob, off)
    assert e.value.code == abi.E_ARG
    # the well-formed batch still runs
    r, _ = gpu_ctx.align_consensus_batch(b.junctions, b.seq_blob, b.seq_off)
    assert r.shape[0] == 4


def test_gather_results_single_rank_equals_fetch(gpu_ctx):
    """dellyhip_gather_results with world = 1 (no RCCL): records, rebased offsets and bytes equal dellyhip_batch_fetch"""
    b = synth.make_batch(200, mode="mixed", n_reads=4, seed=6)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    rb.run(); rb.sync()
    r1, b1 = rb.fetch()
    comm = refine.Comm(gpu_ctx, 0, 1)
    r2, b2, counts = rb.gather(comm, 0)
    comm.close()
    rb.free()
    assert counts.tolist() == [b.n]
    assert r1.tobytes() == r2.tobytes() and b1.tobytes() == b2.tobytes()


def test_rccl_communicator_of_one_rank(gpu_ctx):
    """the RCCL path itself on the one GPU of this box: unique id, ncclCommInitRank(world = 1)"""
    uid = refine.comm_unique_id()
    assert len(uid) == 128
    # world == 1 never loads RCCL; a communicator created WITH an id still must work for the gather
    comm = refine.Comm(gpu_ctx, 0, 1, uid)
    b = synth.make_batch(16, mode="c2", seed=8)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    rb.run()
    r, bl, counts = rb.gather(comm, 0)
    assert r.shape[0] == 16 and counts.tolist() == [16] and int(r["ok"].sum()) >= 14
    comm.close()
    rb.free()


def test_parked_memory_is_reused_and_can_be_returned():
    """the library parks released device / pinned blocks (include/dellyhip.h, memory policy): a second stream of the same
    shape allocates nothing new, dellyhip_trim_memory hands everything back"""
    b = synth.make_batch(400, mode="c2", seed=11)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    ctx.trim_memory()

    def lap():
        st = refine.Stream(ctx, depth=2)
        st.submit(b)
        st.submit(b)
        r0 = st.collect()[0]
        r1 = st.collect()[0]
        st.close()
        return r0, r1

    r0, r1 = lap()
    after_first = _free_bytes()
    r2, r3 = lap()
    after_second = _free_bytes()
    assert after_first - after_second < (1 << 20), "the second stream allocated %d new bytes" % (after_first - after_second)
    for r in (r1, r2, r3):
        assert all(np.array_equal(r0[f], r[f]) for f in r0.dtype.names)
    released = ctx.trim_memory()
    assert released > (64 << 20)              # (two slots: scratch areas, output blocks, staging arenas)
    assert _free_bytes() - after_second > (64 << 20)
    assert ctx.trim_memory() == 0
    r4, _ = ctx.refine(b)                     # and the library works on after a trim
    assert all(np.array_equal(r0[f], r4[f]) for f in r0.dtype.names)
    ctx.close()


def test_parked_memory_is_capped(monkeypatch):
    """DELLYHIP_POOL_LIMIT_MB: beyond the cap the oldest parked blocks go back to the runtime at once"""
    b = synth.make_batch(400, mode="c2", seed=12)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    ctx.trim_memory()
    base = _free_bytes()
    monkeypatch.setenv("DELLYHIP_POOL_LIMIT_MB", "32")
    st = refine.Stream(ctx, depth=2)
    st.submit(b)
    r0 = st.collect()[0]
    held = base - _free_bytes()
    assert held > (256 << 20)                 # two slots' scratch areas alone are ~1.9 GB
    st.close()
    assert base - _free_bytes() < (40 << 20), "parked beyond the cap: %d bytes" % (base - _free_bytes())
    monkeypatch.delenv("DELLYHIP_POOL_LIMIT_MB")
    r1, _ = ctx.refine(b)
    assert all(np.array_equal(r0[f], r1[f]) for f in r0.dtype.names)
    ctx.trim_memory()
    ctx.close()


def test_fetch_into_a_pinned_shared_memory_segment(gpu_ctx):
    """the no-collective return path of bench.py --gpus N: dellyhip_host_register over a POSIX shared-memory segment,
    dellyhip_batch_fetch straight into it, a second mapping of the segment (the merging process) reads the same records"""
    from delly_amd import shmreturn
    b = synth.make_batch(500, mode="mixed", seed=4)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    rb.run(); rb.sync()
    want_r, want_b = rb.fetch()
    rbytes = abi.result_dtype().itemsize
    seg = shmreturn.Segment("pytest_%d" % os.getpid(), 0, 600, rbytes, 500 * 3100 + 64, create=True)
    seg.pin(gpu_ctx)
    for lap in range(2):
        seg.begin()
        used = rb.fetch_into(seg.records_view(), seg.blob_view())
        seg.commit(rb.n, used)
    reader = shmreturn.Segment("pytest_%d" % os.getpid(), 0, 600, rbytes, 500 * 3100 + 64, create=False)
    seqno, rec, blob = reader.read(abi.result_dtype())
    assert seqno == 2 and rec.shape[0] == 500
    assert all(np.array_equal(rec[f], want_r[f]) for f in want_r.dtype.names)
    assert blob.tobytes() == want_b.tobytes()
    del rec, blob
    reader.close()
    rb.free()
    seg.close()


def test_a_failed_call_does_not_wedge_the_host_buffer_entry_points(monkeypatch):
    """the synchronous entry points run through a cached one-slot stream; a collect (or a submit) that fails
    after work was enqueued used to leave the slot 'submitted' for ever -- every later call on the context failed with
    'every slot is in flight or held'.  Now the failed batch is dropped and the next call works."""
    b = synth.make_batch(300, mode="c2", seed=21)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    want, wblob = ctx.refine(b)                       # (creates the cached stream of this mode without injection)
    ctx.close()
    for var in ("DELLYHIP_TEST_FAIL_COLLECT", "DELLYHIP_TEST_FAIL_SUBMIT"):
        monkeypatch.setenv(var, "2")
        ctx = refine.Context()
        ctx.set_chromosomes(b.chroms)
        for k in range(2):
            with pytest.raises(refine.DellyHipError) as e:
                ctx.refine(b)
            assert e.value.code == abi.E_NOMEM and "injected" in str(e.value)
        got, gblob = ctx.refine(b)
        assert all(np.array_equal(got[f], want[f]) for f in want.dtype.names) and gblob.tobytes() == wblob.tobytes()
        monkeypatch.delenv(var)
        ctx.close()


def test_a_failed_collect_drops_one_batch_of_a_pipelined_stream(monkeypatch):
    bs = [synth.make_batch(200, mode="c2", seed=30 + k) for k in range(4)]
    chroms = bs[0].chroms
    ctx = refine.Context()
    wants = []
    for b in bs:
        ctx.set_chromosomes(b.chroms)
        wants.append(ctx.refine(b)[0])
    monkeypatch.setenv("DELLYHIP_TEST_FAIL_COLLECT", "1")
    # the batches carry their own chromosomes: one genome for the stream
    import bench
    chroms, batches = bench.one_genome(synth, bs)
    ctx.set_chromosomes(chroms)
    st = refine.Stream(ctx, depth=3)
    monkeypatch.delenv("DELLYHIP_TEST_FAIL_COLLECT")
    st.submit(batches[0], tag=0); st.submit(batches[1], tag=1)
    with pytest.raises(refine.DellyHipError):
        st.collect()
    assert st.pending() == 1                          # batch 0 is gone, batch 1 still in flight
    st.submit(batches[2], tag=2)
    for k in (1, 2):
        res, blob, tag = st.collect()[:3]
        assert tag == k and np.array_equal(res["ok"], wants[k]["ok"]) and np.array_equal(res["hom_len"], wants[k]["hom_len"])
    st.close()
    ctx.close()
