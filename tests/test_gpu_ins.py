"""-m gpu parity tests of the edlib-equivalent device code behind the insertion path
(splitAlign, src/split.h:480-538), through the C-ABI entry dellyhip_edlib_align:
vectors produced by the reference's vendored edlib (tests/golden/edlib.npz) and the
C restatement on fresh seeds.  Integer / byte outputs: bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_edlib_align_reproduces_reference_golden_vectors(gpu_ctx):
    g = np.load(os.path.join(GOLD, "edlib.npz"), allow_pickle=True)
    n = 0
    for q, t, mode, out, ops in zip(g["q"], g["t"], g["mode"], g["out"], g["ops"]):
        r = gpu_ctx.edlib_align(q, t, int(mode), 2)
        assert tuple(r[:4]) == tuple(int(x) for x in out), (len(q), len(t), int(mode), r[:4], out)
        assert r[4] == ops, (len(q), len(t), int(mode))
        n += 1
    assert n >= 700


def test_edlib_align_tasks_and_edges_vs_port(gpu_ctx, port):
    rng = np.random.default_rng(5)
    cases = [(b"ACGT", b""), (b"", b"ACGT"), (b"A", b"A"), (b"A", b"C"), (b"ACGTACGT" * 8, b"ACGTACGT" * 8),
             (b"AC" * 64, b"GT" * 100), (b"A" * 128, b"A" * 319), (b"ACGT" * 500, b"TTTT" + b"ACGT" * 70)]
    for _ in range(40):
        t = bytes(rng.choice(list(b"ACGTN"), int(rng.integers(1, 320))).astype(np.uint8))
        a = int(rng.integers(0, len(t)))
        q = bytearray(t[a:a + int(rng.integers(1, 256))])
        for k in range(len(q)):
            if rng.random() < 0.08:
                q[k] = rng.choice(list(b"ACGT"))
        cases.append((bytes(q), t))
    for q, t in cases:
        for mode in (0, 1, 2):
            for task in (0, 1, 2):
                want = port.edlib_align(q, t, mode, task)
                got = gpu_ctx.edlib_align(q, t, mode, task)
                assert want is not None
                if task == 0:  # DISTANCE: edlib leaves startLocations NULL
                    assert got[:3] == want[:3] and got[4] == b"" == want[4], (len(q), len(t), mode, task, got, want)
                else:
                    assert got == want, (len(q), len(t), mode, task, got[:4], want[:4])


def test_edlib_align_beyond_the_insertion_kernel_shapes_vs_reference(gpu_ctx, reference):
    """dellyhip_edlib_align(HW / SHW / PATH ...) beyond 319 x 2 048 goes through the strip machinery of
    dellyhip_edlib_align_full instead of failing (the reference's edlibAlign has no limit)"""
    rng = np.random.default_rng(8)
    cases = [(b"ACGT", b"A" * 400)]
    for tn, qn in ((400, 120), (1500, 900), (5000, 1500), (900, 2600)):
        t = bytes(rng.choice(list(b"ACGT"), tn).astype(np.uint8))
        a = int(rng.integers(0, max(1, tn - min(qn, tn))))
        q = bytearray((t[a:a + qn] + bytes(rng.choice(list(b"ACGT"), max(0, qn - (tn - a))).astype(np.uint8)))[:qn])
        for k in range(len(q)):
            if rng.random() < 0.05:
                q[k] = rng.choice(list(b"ACGT"))
        cases.append((bytes(q), t))
    for q, t in cases:
        for mode in (0, 1, 2):
            for task in (0, 2):
                want = reference.edlib_align(q, t, mode, task)
                got = gpu_ctx.edlib_align(q, t, mode, task)
                assert want is not None
                if task == 0:
                    assert got[:3] == want[:3], (len(q), len(t), mode, task, got[:4], want[:4])
                else:
                    assert got == want, (len(q), len(t), mode, task, got[:4], want[:4])


def test_nw_distance_bitvector_long_strings(gpu_ctx, port):
    """edlibAlign(NW, DISTANCE) on long-read sized strings (Myers bit-vector kernel): golden vectors of the
    reference's edlib, plus word / lane boundary lengths against the C restatement"""
    g = np.load(os.path.join(GOLD, "longread.npz"), allow_pickle=True)
    n = 0
    for q, t, mode, out in zip(g["q"], g["t"], g["mode"], g["out"]):
        if int(mode) != 0:
            continue
        r = gpu_ctx.edlib_align(q, t, 0, 0)
        assert r[0] == int(out[0]), (len(q), len(t))
        n += 1
    assert n >= 5
    rng = np.random.default_rng(23)
    for ln in (1, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 4095, 4097, 6144):
        t = bytes(rng.choice(list(b"ACGTNacgtR"), ln, p=[.24, .24, .24, .24, .01, .01, .005, .005, .005, .005]).astype(np.uint8))
        q = bytearray(t)
        for k in range(0, len(q), 7):
            if rng.random() < 0.3:
                q[k] = rng.choice(list(b"ACGT"))
        q = bytes(q[: max(1, ln - int(rng.integers(0, 5)))]) + b"ACGT" * int(rng.integers(0, 3))
        want = port.edlib_align(q, t, 0, 0)
        got = gpu_ctx.edlib_align(q, t, 0, 0)
        assert got[0] == want[0], (ln, got[0], want[0])
