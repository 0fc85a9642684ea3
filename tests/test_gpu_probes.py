"""-m gpu parity of probe generation (the per-SV body of _generateProbes, src/coverage.h:196-258): HIP vs the
reference-generated golden probes and vs the C restatement, through the C-ABI; and the device pipeline refinement batch ->
probes -> read classifier against the same chain on the CPU."""
import os

import numpy as np
import pytest

import fuzz
from delly_amd import abi, refine, synth
from util import compare_probes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "probes.npz")


@pytest.mark.parametrize("label", ["c2", "mixed", "ins"])
def test_probes_reproduce_reference_vectors(label):
    z = np.load(GOLD)
    b = synth.make_batch(int(z[label + "_n"]), **eval(str(z[label + "_kwargs"])))
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    rec, blob = ctx.generate_probes(b)
    ctx.close()
    compare_probes(rec, blob, z[label + "_rec"], z[label + "_blob"], label)
    assert int(rec["ok"].sum()) > 60


@pytest.mark.parametrize("mode", ["c2", "mixed", "ins"])
@pytest.mark.parametrize("pi", [0, 1, 3])
def test_probes_vs_port_fuzz(port, mode, pi):
    p = fuzz.params_of(pi)
    b = fuzz.perturbed(200, 9 + pi, mode)
    ctx = refine.Context(params=p)
    ctx.set_chromosomes(b.chroms)
    rec, blob = ctx.generate_probes(b)
    ctx.close()
    want, wblob = port.generate_probes(b, params=p)
    compare_probes(rec, blob, want, wblob, "%s/%d" % (mode, pi))


def test_probes_short_consensus_and_limits(port):
    """consensus shorter than alignConsensus' length test (src/split.h:647, absent from _generateProbes); long-read shapes
    are flagged, never answered"""
    b = synth.make_batch(12, mode="c2", cons_flank=12)     # 24 bp < 2 * 13
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    rec, blob = ctx.generate_probes(b)
    want, wblob = port.generate_probes(b)
    compare_probes(rec, blob, want, wblob, "short")
    lr = synth.make_batch(2, mode="lr", sub_rate=0.01)
    ctx.set_chromosomes(lr.chroms)
    rec, _ = ctx.generate_probes(lr)
    ctx.close()
    assert (rec["ok"] == 0).all() and (rec["status"] == abi.E_LIMIT).all()


def test_refine_probes_classify_pipeline(port):
    """refinement batch -> probes (cut on the device from the batch's descriptors) -> classifier jobs built from those probes"""
    b = synth.make_batch(64, mode="c2", seed=8)
    ctx = refine.Context()
    ctx.set_chromosomes(b.chroms)
    rec, blob = ctx.generate_probes(b)
    want, wblob = port.generate_probes(b)
    compare_probes(rec, blob, want, wblob, "pipeline")
    # reads: the consensus itself (ALT) and the reference around svStart (REF), against bpPoint 0 probes
    ok = np.nonzero(rec["ok"])[0]
    parts, jobs, pos = [blob], [], blob.size
    for k in ok:
        cons = np.frombuffer(b.seqs_of(int(k))[0], dtype=np.uint8)
        s = int(b.junctions[k]["sv_start"])
        refread = np.char.upper(b.chroms[int(b.junctions[k]["chr"])][s - 75:s + 75].tobytes()).tobytes()
        for read in (cons, np.frombuffer(refread, dtype=np.uint8)):
            parts.append(read)
            jobs.append((int(rec["cons_off0"][k]), int(rec["ref_off0"][k]), pos, int(rec["cons_len0"][k]), int(rec["ref_len0"][k]),
                         read.size, 0, int(k), 60))
            pos += read.size
    J = np.zeros(len(jobs), dtype=abi.align_job_dtype())
    for i, name in enumerate(("cons_off", "ref_off", "seq_off", "cons_len", "ref_len", "seq_len", "file_index", "sv_id", "qual")):
        J[name] = [x[i] for x in jobs]
    allblob = np.concatenate(parts)
    got = ctx.classify_reads(J, allblob)
    ctx.close()
    want = port.classify_reads(J, allblob)
    assert got.tobytes() == want.tobytes()
    # the consensus read supports ALT, the reference read supports REF
    assert (got["type"][0::2] == ord("A")).mean() > 0.9 and (got["type"][1::2] == ord("R")).mean() > 0.9
