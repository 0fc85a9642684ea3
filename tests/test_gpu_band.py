"""-m gpu: the pairwise stages of the long-read consensus on the paths round 6 added -- banded bit-vector distances with several
pairs per wavefront (myers_band.hpp) and msaWfa's diagonal seeding with its 7-mer table in LDS (wfa_seed_kernel) -- on read
sets chosen for the places where those paths hand over to the old ones: letters outside A, C, G, T (the band declines), lengths
further apart than the band, reads so noisy that the distance exceeds the band (not certified: full pass), identical reads,
reads shorter than one block, two reads only, reads beyond the LDS histogram.  The consensus depends on the whole distance
matrix (medoid, order, 80 % cut: src/assemble.h:397-424, :576-596), so loop bodies are compared with the reference itself."""
import os

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 32)


def _ont(rng, s, rate):
    return bytes(synth._ont(rng, np.frombuffer(s, dtype=np.uint8), rate))


def _read_sets(rng, ins):
    """(name, reads) -- every set around one locus; ins: the reads carry an insertion (svt 4: msaWfa)"""
    L = 2300
    base = bytes(synth.ACGT[rng.integers(0, 4, L + 200)])
    if ins:
        base = base[:L // 2] + bytes(synth.ACGT[rng.integers(0, 4, 400)]) + base[L // 2:]

    def cut(rate=0.06, lo=0, hi=60):
        return _ont(rng, base[int(rng.integers(lo, hi + 1)):len(base) - int(rng.integers(lo, hi + 1))], rate)
    sets = []
    sets.append(("plain", [cut() for _ in range(9)]))
    sets.append(("letters N and lower case", [bytes((ord("N") if k % 97 == 13 else (c + 32 if k % 211 == 5 else c)) for k, c in enumerate(cut())) if q in (2, 5) else cut()
                                               for q in range(8)]))
    sets.append(("one read half as long", [cut() for _ in range(6)] + [cut()[:1100]]))
    sets.append(("very noisy reads", [cut(rate=0.16) for _ in range(3)] + [cut() for _ in range(5)]))   # (16 % + 6 % of 2.5 kb: beyond the band of 15.6 % + 32; 25 % runs into msaWfa's documented column limit)
    same = cut()
    sets.append(("identical reads", [same, same[:], same[:], cut(), cut()]))
    sets.append(("two reads", [cut(), cut()]))
    short = bytes(synth.ACGT[rng.integers(0, 4, 160)])
    sets.append(("reads shorter than a block of the band", [_ont(rng, short[int(rng.integers(0, 8)):], 0.05) for _ in range(6)]))
    sets.append(("overhangs beyond the band", [cut(lo=0, hi=450) for _ in range(7)]))
    return sets


def _batch(rng, sets, svt):
    G = synth.ACGT[rng.integers(0, 4, 4000 * len(sets) + 12000)].copy()
    junc = np.zeros(len(sets), dtype=abi.junction_dtype())
    seqs, first = [], 0
    for k, (_, reads) in enumerate(sets):
        s = 3000 + 4000 * k
        junc[k]["svid"] = k
        junc[k]["svt"] = svt
        junc[k]["sv_start"] = s
        junc[k]["sv_end"] = s + (2 if svt == 4 else 1200)
        junc[k]["ins_len"] = 400 if svt == 4 else 0
        junc[k]["n_seq"] = len(reads)
        junc[k]["seq_first"] = first
        first += len(reads)
        seqs += [np.frombuffer(r, dtype=np.uint8) for r in reads]
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs], dtype=np.uint64)
    return synth.Batch([G], junc, np.concatenate(seqs), off, 2, None)


@pytest.mark.parametrize("svt", [2, 4])
def test_pairwise_stage_hand_overs_vs_reference(reference, svt):
    rng = np.random.default_rng(100 + svt)
    sets = _read_sets(rng, ins=(svt == 4))
    # the sets twice over with fresh noise, so that a wavefront's group of pairs mixes certified and declined ones
    sets = sets + _read_sets(rng, ins=(svt == 4))
    b = _batch(rng, sets, svt)
    params = abi.params_lr(realign=True)
    ctx = refine.Context(params=params)
    try:
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=False)
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS, params=params)
        for k, (name, _) in enumerate(sets):
            compare(gr[k:k + 1], gb, rr[k:k + 1], rb, fields=CORE, blobs=("cons", "allele"), label="svt %d, %s" % (svt, name))
        assert int((gr["status"] != 0).sum()) == 0
        assert int((gr["sr_support"] > 0).sum()) >= len(sets) - 2
    finally:
        ctx.close()


def test_band_and_lds_seeding_off_give_the_same_records(monkeypatch):
    """DELLYHIP_MYERS_BAND=0 / DELLYHIP_WFA_LDS_SEED=0 select the round-5 paths: the records must not depend on the choice"""
    b4 = synth.make_batch(24, mode="lrins", n_reads=9, sub_rate=0.06, seed=5)
    b2 = synth.make_batch(24, mode="lr", n_reads=9, sub_rate=0.06, seed=6)
    params = abi.params_lr(realign=True)
    out = {}
    for band, seed in (("1", "1"), ("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("DELLYHIP_MYERS_BAND", band)
        monkeypatch.setenv("DELLYHIP_WFA_LDS_SEED", seed)
        ctx = refine.Context(params=params)
        try:
            res = []
            for b in (b4, b2):
                ctx.set_chromosomes(b.chroms)
                r, bl = ctx.refine(b, want_alignment=False)
                res.append((r.tobytes(), bl.tobytes()))
            out[(band, seed)] = res
        finally:
            ctx.close()
    first = out[("1", "1")]
    for key, res in out.items():
        assert res == first, key
