"""-m gpu parity tests of the long-read shapes (BASELINE config C4: ~2 kb consensus, ~7 kb
window, src/tegua.h:237-241 parameters, alignConsensus(..., realign=true)): the strip kernel
(delly_amd/csrc/lr_kernel.hpp) through the C-ABI against the C restatement and, when oracle/_ref
exists, the reference itself.  Integer / byte outputs: bit-exact."""
import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, INTERNAL_FOUND, compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr_ctx():
    ctx = refine.Context(params=abi.params_lr(realign=True))
    yield ctx
    ctx.close()


def test_lr_align_consensus_vs_port(lr_ctx, port):
    b = synth.make_batch(18, mode="lr", sub_rate=0.01)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL + INTERNAL_FOUND, label="hip-vs-port")
    assert int(gr["ok"].sum()) >= 15


def test_lr_vs_reference(lr_ctx, reference):
    b = synth.make_batch(6, mode="lr", sub_rate=0.01, first=100)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    rr, rb = reference.refine_batch(b, params=abi.params_lr(realign=True))
    compare(gr, gb, rr, rb, label="hip-vs-reference")


def test_lr_without_realign_and_mixed_with_short(port):
    """realign off: reverse-complemented consensus sequences are NOT flipped (split.h:564);
    short-read-shaped junctions in the same batch still take the packed kernels"""
    ctx = refine.Context(params=abi.params_lr(realign=False))
    a = synth.make_batch(6, mode="lr", sub_rate=0.01, first=40)
    ctx.set_chromosomes(a.chroms)
    gr, gb = ctx.refine(a, want_alignment=False)
    pr, pb = port.refine_batch(a, params=abi.params_lr(realign=False), want_alignment=False)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="no-realign")
    ctx.close()


def test_lr_reproduces_reference_golden_vectors(lr_ctx):
    """HIP strip kernel vs the committed outputs of the reference itself (tests/golden/batch_u_lr.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_u_lr.npz"), allow_pickle=True)
    b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    compare(gr, gb, g["results"], g["blob"], label="batch_u_lr.npz")
