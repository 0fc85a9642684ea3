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
    """every field of the record (probes beyond 256 bytes take the one-job-per-wavefront kernel since round 2: no
    DELLYHIP_E_LIMIT below 6144 bytes)"""
    assert (got["status"] == 0).all()
    for f in ("file_index", "sv_id", "dist_alt", "dist_ref", "type", "qual"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, (f, bad[:8], got[f][bad[:8]], want[f][bad[:8]], jobs[bad[:8]])
    return int(got.shape[0])


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


def test_classifier_probes_beyond_256_bytes(reference):
    """probes of 300 .. 900 bytes (long micro-homology): third launch, one job per wavefront"""
    rng = np.random.default_rng(12)
    parts, rows, pos = [], [], 0

    def put(a):
        nonlocal pos
        parts.append(a)
        o = pos
        pos += a.size
        return o
    for k in range(12):
        L = int(rng.integers(260, 900))
        G = synth.ACGT[rng.integers(0, 4, 2 * L + 400)]
        alt = np.concatenate([G[:L + 100], G[L + 300:]])
        cons_probe = alt[100 - 13:100 + L + 13]        # spans the junction of the ALT haplotype
        ref_probe = G[100 - 13:100 + L + 13]
        co, ro = put(cons_probe), put(ref_probe)
        for r in range(6):
            src = alt if r % 2 == 0 else G
            a = int(rng.integers(0, 60))
            read = synth._mutate(rng, src[a:a + L + 200], 0.01)
            so = put(read)
            rows.append((co, ro, so, cons_probe.size, ref_probe.size, read.size, 0, k, int(rng.integers(0, 61))))
    jobs = np.zeros(len(rows), dtype=abi.align_job_dtype())
    for i, r in enumerate(rows):
        for f, v in zip(("cons_off", "ref_off", "seq_off", "cons_len", "ref_len", "seq_len", "file_index", "sv_id", "qual"), r):
            jobs[i][f] = v
    blob = np.concatenate(parts)
    want = reference.classify_reads(jobs, blob)
    ctx = _ctx(0.95)
    got = ctx.classify_reads(jobs, blob)
    ctx.close()
    _same(got, want, jobs)
    assert len(set(bytes(got["type"]).decode())) >= 2
