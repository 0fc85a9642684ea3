"""-m gpu parity of the split-read genotyping classifier (src/coverage.h:412-434, _editDistanceHW
:107-115): HIP vs the reference-generated golden vectors and vs the C restatement, through the C-ABI."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "align_jobs.npz")
FQ = {"plain": 0.95, "weird": 0.95, "lowq": 0.4}


def _same(got, want, jobs):
    """every field of the record; probes beyond the kernel limit must be flagged, never answered"""
    big = np.maximum(jobs["cons_len"], jobs["ref_len"]) > 256
    assert (got["status"][big] == abi.E_LIMIT).all() and (got["type"][big] == ord("N")).all()
    ok = ~big
    assert (got["status"][ok] == 0).all()
    for f in ("file_index", "sv_id", "dist_alt", "dist_ref", "type", "qual"):
        bad = np.nonzero(got[f][ok] != want[f][ok])[0]
        assert bad.size == 0, (f, bad[:8], got[f][ok][bad[:8]], want[f][ok][bad[:8]])
    return int(ok.sum())


def _ctx(fq):
    p = abi.params_sr()
    p.flank_quality = fq
    return refine.Context(params=p)


@pytest.mark.parametrize("label", ["plain", "weird", "lowq"])
def test_classifier_reproduces_reference_vectors(label):
    z = np.load(GOLD)
    jobs, blob, want = z[label + "_jobs"], z[label + "_blob"], z[label + "_results"]
    ctx = _ctx(FQ[label])
    got = ctx.classify_reads(jobs, blob)
    ctx.close()
    assert _same(got, want, jobs) > 500
    assert len(set(bytes(want["type"]).decode())) >= 2


@pytest.mark.parametrize("fq", [0.95, 0.9, 0.5, 0.3])
@pytest.mark.parametrize("weird", [False, True])
def test_classifier_vs_port(port, fq, weird):
    jobs, blob = synth.make_align_jobs(60, 30, seed=int(fq * 100) + weird, weird=weird)
    p = abi.params_sr()
    p.flank_quality = fq
    want = port.classify_reads(jobs, blob, params=p)
    ctx = refine.Context(params=p)
    got = ctx.classify_reads(jobs, blob)
    ctx.close()
    _same(got, want, jobs)


def test_classifier_edges(port):
    """no jobs; a job with all three strings empty; reads shorter than the probe; ragged batch size (not a multiple of 64)"""
    ctx = _ctx(0.95)
    assert ctx.classify_reads(np.zeros(0, dtype=abi.align_job_dtype()), np.zeros(0, dtype=np.uint8)).shape[0] == 0
    blob = np.frombuffer(b"ACGTACGTACGTACGTTTGACCATGACCAGTANNACGT", dtype=np.uint8)
    jobs = np.zeros(5, dtype=abi.align_job_dtype())
    jobs["qual"] = 60
    jobs[1]["cons_len"], jobs[1]["ref_off"], jobs[1]["ref_len"], jobs[1]["seq_off"], jobs[1]["seq_len"] = 30, 4, 30, 0, 7
    jobs[2]["cons_len"], jobs[2]["ref_len"], jobs[2]["seq_len"] = 16, 0, 38
    jobs[3]["cons_off"], jobs[3]["cons_len"], jobs[3]["ref_off"], jobs[3]["ref_len"], jobs[3]["seq_len"] = 16, 16, 20, 14, 38
    jobs[4]["cons_len"], jobs[4]["ref_len"], jobs[4]["seq_len"] = 12, 12, 0
    want = port.classify_reads(jobs, blob)
    got = ctx.classify_reads(jobs, blob)
    _same(got, want, jobs)
    # a bad blob range is an argument error, not a crash
    jobs[0]["seq_off"], jobs[0]["seq_len"] = 30, 100
    with pytest.raises(refine.DellyHipError):
        ctx.classify_reads(jobs, blob)
    ctx.close()


def test_resident_jobs_large_batch(port):
    """one process_batch worth of work kept in HBM: run twice, identical records, spot-checked against the port"""
    jobs, blob = synth.make_align_jobs(400, 40, seed=5)
    ctx = _ctx(0.95)
    rj = refine.ResidentJobs(ctx, jobs, blob)
    rj.run()
    a = rj.fetch()
    rj.run()
    b = rj.fetch()
    ms, launches = rj.kernel_ms()
    assert launches == 2 and ms > 0
    rj.free()
    ctx.close()
    assert a.tobytes() == b.tobytes()
    sel = np.arange(0, jobs.shape[0], 7)
    want = port.classify_reads(jobs[sel], blob)
    _same(a[sel], want, jobs[sel])
