"""Perturbed synthetic junctions for the fuzz parity tests (CPU: restatement vs reference build,
GPU: HIP vs restatement): breakpoint estimates jittered as paired-end estimates are, consensus
sequences trimmed / mutated / with small indels, chromosomes cut so that the reference windows
clip at the chromosome ends, lower-case and N stretches."""
import numpy as np

from delly_amd import abi, synth


def perturbed(n, seed, mode):
    rng = np.random.default_rng(seed * 7919 + 13)
    b = synth.make_batch(n, seed=seed, mode=mode)
    junc = b.junctions.copy()
    seqs = []
    for k in range(n):
        c = np.frombuffer(b.seqs_of(k)[0], dtype=np.uint8).copy()
        what = k % 8
        if what == 1 and c.size > 80:            # trimmed consensus
            a = int(rng.integers(0, c.size // 3)); z = int(rng.integers(0, c.size // 3))
            c = c[a:c.size - z]
        elif what == 2:                           # substitution burst
            p = int(rng.integers(0, max(1, c.size - 12)))
            c[p:p + 12] = synth.ACGT[rng.integers(0, 4, min(12, c.size - p))]
        elif what == 3 and c.size > 40:           # small indels
            p = int(rng.integers(10, c.size - 10))
            c = np.concatenate([c[:p], synth.ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))], c[p:]]) if rng.integers(0, 2) else np.delete(c, slice(p, p + int(rng.integers(1, 4))))
        elif what == 4:                           # lower case / N in the consensus (delly upper-cases reads earlier; the kernels must not care)
            p = int(rng.integers(0, max(1, c.size - 5)))
            c[p] = ord("N")
        if mode == "ins":
            c = c[:319]
        junc[k]["sv_start"] += int(rng.integers(-25, 26)) if what >= 5 else 0
        junc[k]["sv_end"] += int(rng.integers(-25, 26)) if what >= 6 else 0
        if junc[k]["svt"] < 5 and junc[k]["sv_end"] < junc[k]["sv_start"]:
            junc[k]["sv_end"] = junc[k]["sv_start"]
        seqs.append(c)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    return synth.Batch(b.chroms, junc, np.concatenate(seqs), off, 0, None)


def clipped(n, seed, mode):
    """One junction per chromosome pair, the chromosomes cut a few bases around the breakpoints so
    that every reference window is clipped at a chromosome end (src/split.h:60-176 max/min clamps)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        b = synth.make_batch(1, seed=seed, first=k, mode=mode)
        j = b.junctions.copy()
        s, e = int(j[0]["sv_start"]), int(j[0]["sv_end"])
        if len(b.chroms) == 1 or j[0]["chr"] == j[0]["chr2"]:
            c = int(j[0]["chr"])
            lo = max(0, min(s, e) - int(rng.integers(5, 160)))
            hi = min(b.chroms[c].size, max(s, e) + int(rng.integers(5, 160)))
            chroms = list(b.chroms)
            chroms[c] = b.chroms[c][lo:hi].copy()
            j[0]["sv_start"] = s - lo
            j[0]["sv_end"] = e - lo
        else:
            chroms = list(b.chroms)
            for c, f in ((int(j[0]["chr"]), "sv_start"), (int(j[0]["chr2"]), "sv_end")):
                p = int(j[0][f])
                lo = max(0, p - int(rng.integers(5, 160)))
                hi = min(chroms[c].size, p + int(rng.integers(5, 160)))
                chroms[c] = chroms[c][lo:hi].copy()
                j[0][f] = p - lo
        out.append(synth.Batch(chroms, j, b.seq_blob, b.seq_off, 0, None))
    return out


PARAM_SETS = [None,
              (5, -4, -10, -1, 2, 5, 120, 40, 0.8, 0),
              (3, -2, -3, -1, 2, 20, 10000, 100, 0.95, 0),
              (5, -4, -10, -1, 2, 13, 500, 100, 0.9, 0)]


def params_of(i):
    t = PARAM_SETS[i]
    return abi.Params(*t) if t else abi.params_sr()
