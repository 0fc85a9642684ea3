"""-m gpu: edlibAlign at the C++ boundary (SURVEY.md 8b lists it among the signatures to preserve).

dellyhip_edlib_align_full returns everything EdlibAlignResult holds -- editDistance (with the caller's k), ALL optimal end
locations, the start location of each, the alignment of the first pair -- for any shape of the path (short probes, 2 kb
long-read strings in edlib's Hirschberg regime, with or without the 20 extended-IUPAC equalities).  Compared here with the
reference's real edlib (oracle/_ref), first through ctypes, then through a COMPILED caller of
include/delly_dropin/edlib.h (tests/cpp/edlib_dropin_test.cpp: `#include "edlib.h"` resolves to the drop-in)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from delly_amd import refine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "edlib_dropin_test")
IUPAC = b"ACGT-MRWBSYDKEF"


def _mutate(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(list(b"ACGT")))
        if x < rate:
            out.append(rng.choice(list(b"ACGT")))
        else:
            out.append(ch)
    return bytes(out)


def _cases():
    rng = np.random.default_rng(11)
    cases = []   # (k, mode, task, eq, query, target)

    def dna(n, alpha=b"ACGT"):
        return bytes(rng.choice(list(alpha), size=n).tolist())

    # short: probes in reads (HW, k as the genotyper sets it), prefixes, globals; every task; repeats give many end locations
    for i in range(60):
        t = dna(int(rng.integers(20, 400)), b"ACGT" if i % 3 else b"AC")
        a = int(rng.integers(0, max(1, len(t) - 10)))
        q = _mutate(rng, t[a:a + int(rng.integers(5, 60))], 0.08) or b"A"
        for mode in (0, 1, 2):
            cases.append((-1, mode, int(rng.integers(0, 3)), 0, q, t))
        cases.append((int(rng.integers(0, 12)), 2, 0, 0, q, t))          # src/coverage.h:111
        cases.append((int(rng.integers(0, 40)), 0, 0, 0, q, _mutate(rng, q, 0.2) or b"C"))   # src/merge.h:217
    # homopolymer / STR targets: dozens of co-optimal ends, end location -1, start locations of each
    for q, t in ((b"AAAA", b"A" * 70), (b"CACACA", b"CA" * 40), (b"GGGG", b"ACACACAC"), (b"ACGT" * 16, b"ACGT" * 16), (b"A" * 64, b"A" * 200),
                 (b"T", b"T"), (b"T", b"G")):
        for mode in (0, 1, 2):
            for task in (0, 1, 2):
                cases.append((-1, mode, task, 0, q, t))
    # empty operands (src/edlib.cpp:160-178)
    for mode in (0, 1, 2):
        cases.append((-1, mode, 2, 0, b"", b"ACGT"))
        cases.append((3, mode, 1, 0, b"ACGTAC", b""))
    # long-read shapes: NW / HW PATH of ~2 kb strings = edlib's Hirschberg regime; with the IUPAC equalities as msaEdlib /
    # msaWfa call it (src/assemble.h:447,693)
    base = dna(2300)
    r1, r2 = _mutate(rng, base, 0.06), _mutate(rng, base, 0.06)
    cons = bytearray(r2)
    for j in rng.integers(0, len(cons), size=120):
        cons[j] = IUPAC[int(rng.integers(5, 15))]
    cons = bytes(cons)
    cases += [(-1, 0, 2, 0, r1, r2), (-1, 0, 0, 0, r1, r2), (-1, 0, 2, 1, r1, cons), (-1, 2, 2, 1, r1[300:1900], cons), (-1, 2, 1, 0, r1[500:900], r2),
              (-1, 1, 2, 0, r1[:700], r2), (-1, 2, 0, 0, r1[100:160], base), (150, 0, 0, 0, r1, r2), (40, 0, 1, 0, r1, r2),
              (-1, 2, 2, 0, dna(800), dna(6000)), (-1, 0, 2, 1, _mutate(rng, cons, 0.03), cons)]
    return cases


CASES = _cases()


def _full(ctx, q, t, k, mode, task, eq):
    lib = ctx.lib
    qa, ta = np.frombuffer(q, dtype=np.uint8), np.frombuffer(t, dtype=np.uint8)
    cap = len(t) + 1
    ends, starts = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    ops = np.zeros(len(q) + len(t) + 64, dtype=np.uint8)
    ed, nloc, nops = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.dellyhip_edlib_align_full(ctx._ctx, qa.ctypes.data_as(C.c_char_p) if len(q) else None, len(q), ta.ctypes.data_as(C.c_char_p) if len(t) else None,
                                       len(t), int(k), int(mode), int(task), int(eq), C.byref(ed), C.byref(nloc),
                                       ends.ctypes.data_as(C.POINTER(C.c_int32)), starts.ctypes.data_as(C.POINTER(C.c_int32)), cap,
                                       ops.ctypes.data_as(C.POINTER(C.c_ubyte)), int(ops.size), C.byref(nops))
    ctx._check(rc)
    return dict(ed=ed.value, ends=ends[:nloc.value].tolist(), starts=starts[:nloc.value].tolist(), ops=ops[:nops.value].tobytes())


def _expect(reference, k, mode, task, eq, q, t):
    return reference.edlib_align_full(q, t, k=k, mode=mode, task=task, iupac=bool(eq))


def test_c_abi_full_result_vs_the_reference_edlib(gpu_ctx, reference):
    for i, (k, mode, task, eq, q, t) in enumerate(CASES):
        want = _expect(reference, k, mode, task, eq, q, t)
        got = _full(gpu_ctx, q, t, k, mode, task, eq)
        label = "case %d (k %d mode %d task %d eq %d, %d x %d)" % (i, k, mode, task, eq, len(q), len(t))
        assert want["status"] == 0, label
        assert got["ed"] == want["ed"], label
        assert got["ends"] == want["ends"], label
        if want["starts"] is not None:
            assert got["starts"] == want["starts"], label
        assert got["ops"] == want["ops"], label


def test_compiled_caller_of_the_drop_in_edlib_h(tmp_path, reference):
    if not os.path.exists(BIN):
        pytest.skip("tests/cpp/_build/edlib_dropin_test not built (needs /root/reference for the include path)")
    extra = [(-1, 0, 0, 2, b"ACGT", b"ACGT")]          # an equality set the drop-in refuses: EDLIB_STATUS_ERROR, nothing allocated
    cases = CASES + extra
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(cases)))
        for k, mode, task, eq, q, t in cases:
            f.write(struct.pack("<6i", k, mode, task, eq, len(q), len(t)))
            f.write(q)
            f.write(t)
    p = subprocess.run([BIN, "run", fin, fout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = open(fout, "rb").read()
    o = 0
    for i, (k, mode, task, eq, q, t) in enumerate(cases):
        status, ed, nloc, alen, alpha, has_e, has_s, has_a, istart = struct.unpack_from("<9i", d, o)
        o += 36
        label = "case %d (k %d mode %d task %d eq %d, %d x %d)" % (i, k, mode, task, eq, len(q), len(t))
        if eq == 2:
            assert status == 1 and not has_e and not has_s and not has_a, label
            continue
        want = _expect(reference, k, mode, task, eq, q, t)
        assert status == 0 and want["status"] == 0, label
        assert ed == want["ed"] and nloc == len(want["ends"]) and alpha == want["alphabet"], label
        ends = list(struct.unpack_from("<%di" % nloc, d, o)); o += 4 * nloc
        starts = list(struct.unpack_from("<%di" % nloc, d, o)); o += 4 * nloc
        assert ends == want["ends"], label
        assert bool(has_e) == (nloc > 0), label                    # NULL exactly when the reference's is (beyond k)
        assert bool(has_s) == (want["starts"] is not None), label
        if want["starts"] is not None:
            assert starts == want["starts"] and istart == want["starts"][0], label
        ops = b""
        if has_a:
            ops = d[o:o + alen]; o += alen
        assert ops == want["ops"] and bool(has_a) == (task == 2 and ed >= 0 and len(q) > 0 and len(t) > 0), label
        for fmt in (0, 1):
            ln = struct.unpack_from("<i", d, o)[0]; o += 4
            if has_a:
                text = d[o:o + ln]; o += ln
                assert text == reference.edlib_cigar(ops, fmt), label
            else:
                assert ln == -1
    assert struct.unpack_from("<3i", d, o) == (-1, 0, 0)          # edlibDefaultAlignConfig: k = -1, NW, DISTANCE
