"""-m gpu parity tests of the long-read MSA (msaEdlib, src/assemble.h:383-473) and of the
long-read loop body msaEdlib + alignConsensus(realign) through the C-ABI, against vectors of the
reference itself (tests/golden/longread.npz) and the C restatement.  Byte outputs: bit-exact."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lr_ctx():
    ctx = refine.Context(params=abi.params_lr(realign=True))
    yield ctx
    ctx.close()


def test_msa_edlib_reproduces_reference_golden_vectors(lr_ctx):
    g = np.load(os.path.join(GOLD, "longread.npz"), allow_pickle=True)
    for reads, rows, cs in zip(g["msa_sets"], g["msa_rows"], g["msa_cs"]):
        r, c = lr_ctx.msa_edlib(list(reads))
        assert r == int(rows)
        assert c == cs, (len(c), len(cs))


def _ont(rng, s, rate):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(rng.choice(list(b"ACGT")))
            out.append(ch)
            continue
        if u < rate:
            out.append(rng.choice(list(b"ACGT")))
            continue
        out.append(ch)
    return bytes(out)


def test_msa_edlib_vs_port_small_and_edge(lr_ctx, port):
    rng = np.random.default_rng(31)
    old = port.params
    port.params = abi.params_lr()
    try:
        for it in range(8):
            L = int(rng.integers(120, 900))
            base = bytes(rng.choice(list(b"ACGT"), L + 60).astype(np.uint8))
            n = [1, 2, 3, 4, 7, 12, 15, 16][it]
            reads = [_ont(rng, base[int(rng.integers(0, 30)):L + 30 + int(rng.integers(0, 30))], 0.08) for _ in range(n)]
            assert lr_ctx.msa_edlib(reads) == port.msa_edlib(reads), (it, n, L)
    finally:
        port.params = old


def test_refine_batch_lr_reproduces_reference_golden_vectors(lr_ctx):
    """msaEdlib + alignConsensus(realign) vs the committed outputs of the reference (batch_full_lr_n8.npz)"""
    g = np.load(os.path.join(GOLD, "batch_full_lr_n8.npz"), allow_pickle=True)
    b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
    assert b.with_msa == 2
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=True)
    compare(gr, gb, g["results"], g["blob"], label="batch_full_lr_n8.npz")
    assert int(gr["ok"].sum()) == b.n


def test_refine_batch_lr_vs_port(lr_ctx, port):
    b = synth.make_batch(6, mode="lr", n_reads=5, sub_rate=0.05, seed=77, first=20)
    lr_ctx.set_chromosomes(b.chroms)
    gr, gb = lr_ctx.refine(b, want_alignment=False)
    pr, pb = port.refine_batch(b, params=abi.params_lr(realign=True), want_alignment=False)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="hip-vs-port")
