"""-m gpu parity of the batched _editDistanceNW (src/genotype.h:21-30, call sites :276,:284): HIP vs the
reference-generated golden distances and vs the C restatement, through the C-ABI."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "nw_jobs.npz")


def _check(got, want, jobs):
    # (pairs with both strings beyond one bit-vector pass -- 6144 rows -- run in strips since round 2: no size limit)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:8], got[bad[:8]], want[bad[:8]])


@pytest.mark.parametrize("label", ["plain", "weird"])
def test_nw_batch_reproduces_reference_distances(gpu_ctx, label):
    z = np.load(GOLD)
    jobs, blob, want = z[label + "_jobs"], z[label + "_blob"], z[label + "_dist"]
    got = gpu_ctx.edit_distance_nw_batch(jobs, blob)
    _check(got, want, jobs)
    if label == "weird":   # the set holds pairs with both strings > 6144 bytes
        assert (np.minimum(jobs["query_len"], jobs["target_len"]) > 6144).sum() > 0
    assert (got != abi.E_LIMIT).all()


def test_nw_batch_vs_port(gpu_ctx, port):
    jobs, blob = synth.make_nw_jobs(60, seed=77, weird=True, min_half=300, max_half=1400)
    want = port.edit_distance_nw_batch(jobs, blob, n_threads=8)
    _check(gpu_ctx.edit_distance_nw_batch(jobs, blob), want, jobs)


def test_nw_batch_three_word_patterns(gpu_ctx, port):
    """patterns of 4200 .. 6000 rows: three 32-row words per lane"""
    jobs, blob = synth.make_nw_jobs(6, seed=78, min_half=2100, max_half=3000)
    want = port.edit_distance_nw_batch(jobs, blob, n_threads=8)
    _check(gpu_ctx.edit_distance_nw_batch(jobs, blob), want, jobs)
    assert int(jobs["query_len"].min()) > 4096


@pytest.mark.parametrize("lo,hi,n", [(60, 500, 301), (1050, 1500, 120), (100, 1500, 200)])
def test_nw_batch_two_pairs_per_wavefront(gpu_ctx, port, lo, hi, n):
    """strings of 120 .. 1000 bytes (one word per lane for two pairs), 2.1 .. 3 kb (three words) and a mix where only some
    adjacent jobs pay: the side-by-side routine myers_nw_fast_x2, odd job counts, weird bytes (fallback per pair)"""
    jobs, blob = synth.make_nw_jobs(n, seed=91 + lo, weird=(lo == 100), min_half=lo, max_half=hi)
    want = port.edit_distance_nw_batch(jobs, blob, n_threads=8)
    _check(gpu_ctx.edit_distance_nw_batch(jobs, blob), want, jobs)


def test_nw_batch_edges_and_resident(gpu_ctx, port):
    assert gpu_ctx.edit_distance_nw_batch(np.zeros(0, dtype=abi.nw_job_dtype()), np.zeros(0, dtype=np.uint8)).shape[0] == 0
    jobs, blob = synth.make_nw_jobs(300, seed=5)
    rj = refine.ResidentNwJobs(gpu_ctx, jobs, blob)
    rj.run()
    a = rj.fetch()
    rj.run()
    b = rj.fetch()
    ms, launches = rj.kernel_ms()
    rj.free()
    assert launches == 2 and ms > 0 and (a == b).all()
    sel = np.arange(0, jobs.shape[0], 11)
    assert (a[sel] == port.edit_distance_nw_batch(jobs[sel], blob, n_threads=8)).all()
    bad = jobs[:1].copy()
    bad["target_len"] = blob.size + 1
    with pytest.raises(refine.DellyHipError):
        gpu_ctx.edit_distance_nw_batch(bad, blob)
