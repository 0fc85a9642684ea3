"""-m gpu parity tests of the MSA stage (unit U_full) through the C-ABI:
lcs / gotoh / msa single-item wrappers and refine_batch (msa + alignConsensus)
against the golden vectors of the reference and the C restatement."""
import glob
import os

import numpy as np
import pytest

from delly_amd import refine, synth
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def prim():
    return np.load(os.path.join(GOLD, "primitives.npz"), allow_pickle=True)


def test_lcs_golden(gpu_ctx, prim):
    for a, b, o in zip(prim["lcs_a"], prim["lcs_b"], prim["lcs_out"]):
        assert gpu_ctx.lcs(a, b) == o, (a, b)


def test_lcs_foreign_bytes(gpu_ctx, port):
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = bytes(rng.choice(list(b"ACGTNacgtRY"), int(rng.integers(1, 250))).astype(np.uint8))
        b = bytes(rng.choice(list(b"ACGTNacgtRY"), int(rng.integers(1, 300))).astype(np.uint8))
        assert gpu_ctx.lcs(a, b) == port.lcs(a, b)


def test_gotoh_golden(gpu_ctx, prim):
    for a1, a2, sc, rows in zip(prim["gotoh_a1"], prim["gotoh_a2"], prim["gotoh_score"], prim["gotoh_rows"]):
        s2, r2 = gpu_ctx.gotoh(list(a1), list(a2))
        assert s2 == sc
        assert r2 == list(rows)


def test_gotoh_random_profiles(gpu_ctx, port):
    rng = np.random.default_rng(5)
    for it in range(16):
        base = bytes(rng.choice(list(b"ACGT"), 330).astype(np.uint8))

        def var(o, L, rate=0.03):
            x = bytearray(base[o:o + L])
            for k in range(len(x)):
                if rng.random() < rate:
                    x[k] = rng.choice(list(b"ACGTN"))
            if rng.random() < 0.5 and len(x) > 30:
                del x[int(rng.integers(10, len(x) - 10))]
            return bytes(x)

        groups = []
        for g in range(2):
            rows = [var(int(rng.integers(0, 60)), int(rng.integers(100, 250)))]
            for _ in range(int(rng.integers(0, 4))):
                _, rows = port.gotoh(rows, [var(int(rng.integers(0, 60)), int(rng.integers(100, 250)))])
            groups.append(rows)
        ps, pr = port.gotoh(groups[0], groups[1])
        gs, gr = gpu_ctx.gotoh(groups[0], groups[1])
        assert gs == ps, it
        assert gr == pr, it


def test_msa_golden(gpu_ctx, prim):
    for reads, rows, cs in zip(prim["msa_sets"], prim["msa_rows"], prim["msa_cs"]):
        r, c = gpu_ctx.msa(list(reads))
        assert (r, c) == (int(rows), cs)


def test_refine_batch_golden(gpu_ctx):
    n = 0
    for path in sorted(glob.glob(os.path.join(GOLD, "batch_full_*.npz"))):
        g = np.load(path, allow_pickle=True)
        if "lr" in g.files and int(g["lr"]):
            continue  # long-read parameters: tests/test_gpu_lrmsa.py
        b = synth.make_batch(int(g["n"]), **eval(str(g["kwargs"])))
        gpu_ctx.set_chromosomes(b.chroms)
        gr, gb = gpu_ctx.refine(b, want_alignment=True)
        compare(gr, gb, g["results"], g["blob"], label=os.path.basename(path))
        n += b.n
    assert n >= 60


@pytest.mark.parametrize("mode,n_reads,n", [("c2", 20, 64), ("mixed", 9, 96), ("c2", 2, 40), ("c2", 1, 8)])
def test_refine_batch_vs_port(gpu_ctx, port, mode, n_reads, n):
    b = synth.make_batch(n, mode=mode, n_reads=n_reads, seed=99, first=300)
    gpu_ctx.set_chromosomes(b.chroms)
    gr, gb = gpu_ctx.refine(b, want_alignment=True)
    pr, pb = port.refine_batch(b)
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="hip-vs-port")


def test_score_table_and_direct_float_paths_agree(port):
    """msa() uses an int8 table of (type1, type2) profile scores; junctions whose nodes have
    more column types than the table holds (DELLYHIP_MSA_TMAX) or more than 319 columns go to
    the direct-float kernel.  Both routes must give the reference's bytes."""
    from delly_amd import refine
    b = synth.make_batch(48, mode="mixed", n_reads=9, seed=21)
    pr, pb = port.refine_batch(b, want_alignment=False)
    old = os.environ.get("DELLYHIP_MSA_TMAX")
    try:
        for tmax in ("96", "6", "0"):
            os.environ["DELLYHIP_MSA_TMAX"] = tmax
            ctx = refine.Context()
            ctx.set_chromosomes(b.chroms)
            gr, gb = ctx.refine(b, want_alignment=False)
            ctx.close()
            compare(gr, gb, pr, pb, fields=CORE + INTERNAL, blobs=("cons", "allele"), label="tmax=" + tmax)
    finally:
        if old is None:
            os.environ.pop("DELLYHIP_MSA_TMAX", None)
        else:
            os.environ["DELLYHIP_MSA_TMAX"] = old


def test_msa_long_nodes_take_the_direct_kernel(gpu_ctx, port):
    # reads 250 bp spread over 430 bp: alignment nodes exceed 319 columns (K = 6..7)
    rng = np.random.default_rng(17)
    for it in range(6):
        base = bytes(rng.choice(list(b"ACGT"), 460).astype(np.uint8))
        reads = []
        while len(reads) < 7:
            o = int(rng.integers(0, 200))
            x = bytearray(base[o:o + 250])
            for k in range(len(x)):
                if rng.random() < 0.01:
                    x[k] = rng.choice(list(b"ACGT"))
            if bytes(x) not in reads:
                reads.append(bytes(x))
        try:
            got = gpu_ctx.msa(reads)
        except Exception as e:  # consensus longer than the 319-byte output cap is a documented limit
            assert "-4" in str(e)
            continue
        assert got == port.msa(reads), it


@pytest.mark.parametrize("team", [2, 4])
def test_msa_team_of_wavefronts_gives_the_same_consensus(reference, team, monkeypatch):
    """msa_team_kernel<W>: W wavefronts per junction (LCS pairs across the team, merges claimed as their children finish) --
    the schedule must not show in the result: whole refine batches vs the reference itself, team size forced"""
    monkeypatch.setenv("DELLYHIP_MSA_TEAM", str(team))
    ctx = refine.Context()
    try:
        for n_reads, seed in ((20, 5), (7, 6), (2, 7), (3, 8)):
            b = synth.make_batch(96, mode="c2", n_reads=n_reads, seed=seed)
            ctx.set_chromosomes(b.chroms)
            gr, gb = ctx.refine(b, want_alignment=False)
            rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=os.cpu_count() or 1)
            compare(gr, gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label="team %d, %d reads" % (team, n_reads))
    finally:
        ctx.close()
