"""-m gpu parity tests of the split-alignment path (unit U), through the C-ABI:
HIP kernels vs the C restatement (and vs the reference itself when oracle/_ref
was built) on the same seeded batches.  Integer/byte outputs: bit-exact."""
import numpy as np
import pytest

from delly_amd import synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu


def _run(gpu_ctx, batch, want_alignment=True):
    gpu_ctx.set_chromosomes(batch.chroms)
    return gpu_ctx.refine(batch, want_alignment=want_alignment)


def test_dpp_lane_shift_semantics(gpu_ctx):
    # the DP hand-off relies on wave_shr:1 / wave_shl:1 crossing all 64 lanes;
    # a wrong shift would break every alignment below, so one tiny case first
    found, r0, r1 = gpu_ctx.long_needle(b"ACGTACGTTTGACCAGTACGATCGATTTGACA" * 2, b"ACGTACGTTTGACCAGTACGATCGATTTGACA" + b"G" * 40 + b"ACGTACGTTTGACCAGTACGATCGATTTGACA")
    assert found
    assert len(r0) == len(r1)


@pytest.mark.parametrize("mode,n", [("c2", 400), ("mixed", 360)])
def test_align_consensus_vs_port(gpu_ctx, port, mode, n):
    b = synth.make_batch(n, mode=mode)
    gr, gb = _run(gpu_ctx, b)
    pr, pb = port.refine_batch(b)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="hip-vs-port")
    assert int(gr["ok"].sum()) > 0.9 * n


def test_align_consensus_vs_reference(gpu_ctx, reference):
    b = synth.make_batch(240, mode="mixed", first=1000)
    gr, gb = _run(gpu_ctx, b)
    rr, rb = reference.refine_batch(b)
    compare(gr, gb, rr, rb, label="hip-vs-reference")


def test_long_needle_single(gpu_ctx, port):
    rng = np.random.default_rng(7)
    for it in range(40):
        m = int(rng.integers(20, 300))
        n = int(rng.integers(60, 1500))
        ref = rng.choice(list(b"ACGT"), n).astype(np.uint8)
        # consensus = two pieces of ref + noise, sometimes unrelated
        if it % 5 == 4:
            cons = rng.choice(list(b"ACGT"), m).astype(np.uint8)
        else:
            a = int(rng.integers(0, max(1, n // 2 - m // 2)))
            bpos = int(rng.integers(n // 2, max(n // 2 + 1, n - m // 2)))
            cons = np.concatenate([ref[a:a + m // 2], ref[bpos:bpos + (m - m // 2)]])
            cons = cons.copy()
            for k in rng.integers(0, cons.size, max(1, cons.size // 50)):
                cons[k] = rng.choice(list(b"ACGT"))
        f1, a0, a1 = gpu_ctx.long_needle(cons.tobytes(), ref.tobytes())
        f2, b0, b1, _ = port.long_needle(cons.tobytes(), ref.tobytes())
        assert f1 == f2, (it, m, n)
        assert a0 == b0 and a1 == b1, (it, m, n)


def test_edge_characters(gpu_ctx, port):
    # lower case, N runs and IUPAC letters in consensus and reference
    rng = np.random.default_rng(11)
    for it in range(30):
        n = int(rng.integers(200, 900))
        ref = rng.choice(list(b"ACGTNRYacgtn"), n, p=[.2, .2, .2, .2, .05, .02, .02, .03, .03, .03, .01, .01]).astype(np.uint8)
        m = int(rng.integers(40, 200))
        a = int(rng.integers(0, n // 3))
        bpos = int(rng.integers(n // 2, n - m // 2 - 1))
        cons = np.concatenate([ref[a:a + m // 2], ref[bpos:bpos + (m - m // 2)]]).copy()
        f1, a0, a1 = gpu_ctx.long_needle(cons.tobytes(), ref.tobytes())
        f2, b0, b1, _ = port.long_needle(cons.tobytes(), ref.tobytes())
        assert (f1, a0, a1) == (f2, b0, b1), it
