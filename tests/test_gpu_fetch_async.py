"""-m gpu: dellyhip_batch_fetch_begin / _end -- the return of a resident batch's results queued behind its kernels and written by a
kernel into the caller's pinned host memory (the pipelined N > 1 step of bench.py) -- must hand out exactly what the blocking
dellyhip_batch_fetch hands out: records with rebased blob offsets and the compacted bytes.  The batches themselves are compared
with oracle/_ref in the other suites; here one of them is, through the asynchronous path."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, shmreturn, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
RB = abi.result_dtype().itemsize
THREADS = os.cpu_count() or 1


def _pinned(ctx, nbytes):
    a = np.zeros(nbytes + 64, dtype=np.uint8)
    off = (-a.ctypes.data) % 64
    v = a[off:off + nbytes]
    ctx.host_register(v.ctypes.data, v.nbytes)
    return a, v


def _same(rec_bytes, blob, used, want_r, want_b):
    rec = rec_bytes[:want_r.shape[0] * RB].view(abi.result_dtype())
    for f in want_r.dtype.names:
        x, y = rec[f], want_r[f]
        ok = (x == y) | ((x != x) & (y != y)) if x.dtype.kind == "f" else (x == y)
        assert ok.all(), f
    assert used == want_b.nbytes and blob[:used].tobytes() == want_b.tobytes()


@pytest.mark.parametrize("mode,n", [("c2", 3000), ("mixed", 700), ("ins", 300)])
def test_async_fetch_equals_blocking_fetch(gpu_ctx, mode, n):
    b = synth.make_batch(n, mode=mode, seed=11)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    rb.run(); rb.sync()
    want_r, want_b = rb.fetch()
    keep_r, rec = _pinned(gpu_ctx, n * RB)
    keep_b, blob = _pinned(gpu_ctx, n * 3100 + 4096)
    for lap in range(3):               # lap 0: batch finished; laps 1, 2: queued straight behind a run, nothing waited for in between
        rec[:] = 0xEE; blob[:] = 0xEE
        if lap:
            rb.run()
        rb.fetch_begin(rec, blob)
        used = rb.fetch_end()
        _same(rec, blob, used, want_r, want_b)
        assert (blob[used:used + 64] == 0xEE).all()          # nothing written beyond the used bytes
    rb.free()
    gpu_ctx.host_unregister(rec.ctypes.data); gpu_ctx.host_unregister(blob.ctypes.data)


@pytest.mark.parametrize("shift", [1, 4, 9])
def test_async_fetch_into_a_destination_that_is_not_16_byte_aligned(gpu_ctx, shift):
    """an unaligned out_blob takes a byte-wise head, 16-byte pieces, a byte-wise tail -- same bytes"""
    n = 1200
    b = synth.make_batch(n, mode="c2", seed=13)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    rb.run(); rb.sync()
    want_r, want_b = rb.fetch()
    keep_r, rec = _pinned(gpu_ctx, n * RB)
    keep_b, blob0 = _pinned(gpu_ctx, n * 3100 + 4096)
    blob = blob0[shift:]
    rec[:] = 0xEE; blob0[:] = 0xEE
    rb.fetch_begin(rec, blob)
    used = rb.fetch_end()
    _same(rec, blob, used, want_r, want_b)
    assert (blob0[:shift] == 0xEE).all() and (blob[used:used + 64] == 0xEE).all()
    rb.free()
    gpu_ctx.host_unregister(rec.ctypes.data); gpu_ctx.host_unregister(blob0.ctypes.data)


def test_async_fetch_vs_reference_through_a_segment_with_the_next_runs_queued(gpu_ctx, reference):
    """the step of bench.py --gpus N: two resident batches on two contexts / streams; run(k), publish the return of k - 2, queue the
    return of k - 1 -- what the segment holds after every step is the reference's answer for that batch"""
    raw = [synth.make_batch(1500, mode="c2", first=k * 1500) for k in range(2)]
    import bench
    chroms, batches = bench.one_genome(synth, raw)
    gpu_ctx.set_chromosomes(chroms)
    other = refine.Context(share_with=gpu_ctx)
    ctxs, streams = [gpu_ctx, other], gpu_ctx.compute_streams()
    rbs = [ctxs[k].upload(b) for k, b in enumerate(batches)]
    want = [reference.refine_batch(b, want_alignment=False, n_threads=THREADS) for b in batches]
    seg = shmreturn.Segment("pytest_async_%d" % os.getpid(), 0, 1600, RB, 1500 * 1400 + (1 << 20), create=True)
    seg.pin(gpu_ctx)
    reader = shmreturn.Segment("pytest_async_%d" % os.getpid(), 0, 1600, RB, 1500 * 1400 + (1 << 20), create=False)
    flying = None
    checked = 0
    for k in range(7):
        rbs[k % 2].run(streams[k % 2])
        if flying is not None:
            used = rbs[flying].fetch_end()
            seg.commit(rbs[flying].n, used)
            seqno, rec, blob = reader.read(abi.result_dtype(), copy=True)
            compare(rec, blob, want[flying][0], want[flying][1], fields=CORE, blobs=("cons", "allele"), label="step %d returns batch %d" % (k, flying))
            checked += 1
            flying = None
        if k > 0:
            seg.begin()
            assert reader.read(abi.result_dtype()) is None       # (the seqlock is odd while a return is in flight)
            rbs[(k + 1) % 2].fetch_begin(seg.records_view(), seg.blob_view())
            flying = (k + 1) % 2
    rbs[flying].fetch_end()
    assert checked == 5
    for x in rbs:
        x.free()
    reader.close()
    seg.close()
    other.close()


def test_async_fetch_argument_errors(gpu_ctx):
    b = synth.make_batch(200, mode="c2", seed=5)
    gpu_ctx.set_chromosomes(b.chroms)
    rb = gpu_ctx.upload(b)
    pageable_r, pageable_b = np.zeros(200 * RB, dtype=np.uint8), np.zeros(200 * 3100, dtype=np.uint8)
    keep_r, rec = _pinned(gpu_ctx, 200 * RB)
    keep_b, blob = _pinned(gpu_ctx, 200 * 3100)
    with pytest.raises(refine.DellyHipError) as e:       # never run
        rb.fetch_begin(rec, blob)
    assert e.value.code == abi.E_ARG
    with pytest.raises(refine.DellyHipError) as e:       # nothing in flight
        rb.fetch_end()
    assert e.value.code == abi.E_ARG
    rb.run()
    with pytest.raises(refine.DellyHipError) as e:       # pageable memory: a kernel cannot write into it
        rb.fetch_begin(pageable_r, pageable_b)
    assert e.value.code == abi.E_ARG and "pinned" in str(e.value)
    want_r, want_b = rb.fetch()
    rb.fetch_begin(rec, blob[:1000])                     # too small for the bytes: the records arrive, the size needed is reported
    with pytest.raises(refine.DellyHipError) as e:       # one fetch per batch in flight
        rb.fetch_begin(rec, blob)
    assert e.value.code == abi.E_ARG and "in flight" in str(e.value)
    with pytest.raises(refine.DellyHipError) as e:
        rb.fetch_end()
    assert e.value.code == abi.E_ARG and "too small" in str(e.value) and e.value.blob_bytes_needed == want_b.nbytes
    rb.fetch_begin(rec, blob)                            # ... and the batch is usable afterwards
    _same(rec, blob, rb.fetch_end(), want_r, want_b)
    rb.fetch_begin(rec, blob)                            # freed with a fetch in flight: waits, no crash
    rb.free()
    gpu_ctx.host_unregister(rec.ctypes.data); gpu_ctx.host_unregister(blob.ctypes.data)
