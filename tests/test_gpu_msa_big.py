"""-m gpu: short-read msa() beyond the standard kernel instance (reads > 256 bytes, > 32 reads, alignment nodes > 512
columns -> dh::msa_big: LCS masks in HBM, direct-float profile Gotoh in strips of 512 rows) and the consensus sequences
beyond 319 bp it produces (split alignment in the strip kernel), bit-compared with the reference (oracle/_ref)."""
import os

import numpy as np
import pytest

from delly_amd import abi, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
THREADS = min(32, os.cpu_count() or 1)


@pytest.mark.parametrize("n,n_reads,read_len", [(24, 12, 300),    # 2 x 300 bp reads: longer than the 256-byte LCS words
                                                (16, 48, 150),    # -p 48: more reads than the standard instance holds
                                                (24, 20, 250),    # fits the standard instance unless a node passes 512 columns
                                                (8, 8, 600),      # nodes of ~1100 columns: three Gotoh strips
                                                (6, 64, 160)])    # the msa_big read cap
def test_refine_batch_big_msa_shapes_vs_reference(gpu_ctx, reference, n, n_reads, read_len):
    b = synth.make_batch(n, mode="c2", n_reads=n_reads, read_len=read_len, cons_flank=read_len, seed=31)
    gpu_ctx.set_chromosomes(b.chroms)
    gr, gb = gpu_ctx.refine(b, want_alignment=True)
    rr, rb = reference.refine_batch(b, n_threads=THREADS)
    compare(gr, gb, rr, rb, fields=CORE, label="msa_big %d x %d" % (n_reads, read_len))
    assert int((gr["status"] != 0).sum()) == 0
    assert int(gr["ok"].sum()) >= n - 2
    assert int(gr["sr_support"].min()) >= min(n_reads, 2)


def test_msa_single_wrapper_big_shapes_vs_reference(gpu_ctx, reference):
    rng = np.random.default_rng(41)
    for n_reads, read_len in ((5, 400), (40, 120), (3, 1000), (64, 100)):
        base = synth.ACGT[rng.integers(0, 4, 2 * read_len)]
        reads = []
        while len(reads) < n_reads:
            o = int(rng.integers(0, read_len))
            r = bytes(synth._mutate(rng, base[o:o + read_len], 0.01))
            if r not in reads:
                reads.append(r)
        assert gpu_ctx.msa(reads) == reference.msa(reads), (n_reads, read_len)
    a, c = bytes(synth.ACGT[rng.integers(0, 4, 900)]), bytes(synth.ACGT[rng.integers(0, 4, 700)])
    assert gpu_ctx.lcs(a, c) == reference.lcs(a, c)


def test_gotoh_single_wrapper_wide_nodes_vs_reference(gpu_ctx, reference):
    rng = np.random.default_rng(43)
    base = synth.ACGT[rng.integers(0, 4, 1500)]
    for r1, m, r2, n in ((1, 700, 1, 650), (3, 900, 2, 800), (40, 200, 1, 180)):
        def node(rows, cols, off):
            out = np.empty((rows, cols), dtype=np.uint8)
            for i in range(rows):
                out[i] = synth._mutate(rng, base[off:off + cols], 0.03)
                g = rng.integers(0, cols, 3)
                out[i, g] = ord("-")
            return out
        a1, a2 = node(r1, m, 10), node(r2, n, 40)
        got = gpu_ctx.gotoh([bytes(x) for x in a1], [bytes(x) for x in a2])
        want = reference.gotoh([bytes(x) for x in a1], [bytes(x) for x in a2])
        assert got == want, (r1, m, r2, n)
