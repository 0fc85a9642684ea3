"""-m gpu: a COMPILED C++ caller of the C-ABI -- tests/cpp/dropin_test.cpp includes the reference's tags.h / align.h and
the drop-in headers include/delly_dropin/{msa,needle,gotoh,split,assemble_msa}.h (torali:: template signatures of
SURVEY.md 8b), drives msa() + alignConsensus() per junction, torali::refineBatch() per chromosome, msaEdlib / msaWfa and
the primitives on seeded batches, and writes the StructuralVariantRecord fields a Delly caller would see.  Compared
here, field by field, with the reference itself (oracle/_ref)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from delly_amd import abi, synth
import pyoracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "dropin_test")


def _write_batch(path, b, params, realign, mode):
    with open(path, "wb") as f:
        f.write(bytes(params))
        f.write(struct.pack("<iii", int(realign), int(mode), len(b.chroms)))
        for c in b.chroms:
            f.write(struct.pack("<Q", c.size))
            f.write(c.tobytes())
        f.write(struct.pack("<i", b.n))
        f.write(np.ascontiguousarray(b.junctions).tobytes())
        f.write(struct.pack("<Q", b.n_seq))
        f.write(np.ascontiguousarray(b.seq_off, dtype=np.uint64).tobytes())
        f.write(b.seq_blob.tobytes())


class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def s(self):
        n = self.take("I")
        v = self.d[self.o:self.o + n]
        self.o += n
        return v

    def sv(self):
        ok, rows = self.take("ii")
        svS, svE, cpl, cph, cel, ceh, ins, cbp, hom, sup = self.take("10i")
        q = self.take("f")
        precise = self.take("i")
        return dict(ok=ok, rows=rows, svStart=svS, svEnd=svE, ciposlow=cpl, ciposhigh=cph, ciendlow=cel, ciendhigh=ceh,
                    insLen=ins, consBp=cbp, homLen=hom, srSupport=sup, q=q, precise=precise, consensus=self.s(), alleles=self.s())


def _run(tmp_path, b, params, realign, mode):
    if not os.path.exists(BIN):
        pytest.fail("tests/cpp/_build/dropin_test is not built (__graft_entry__.build() compiles it where /root/reference exists)")
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_batch(inp, b, params, realign, mode)
    r = subprocess.run([BIN, "run", inp, outp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return _Reader(open(outp, "rb").read())


def _check_sv(got, J, rr, rb, label, with_cons=True):
    assert got["ok"] == int(rr["ok"]), label
    cons = pyoracle.blob_field(rr, rb, "cons")
    if with_cons:
        assert got["consensus"] == cons, label
    if rr["ok"]:
        assert (got["svStart"], got["svEnd"]) == (int(rr["sv_start"]), int(rr["sv_end"])), label
        w = int(rr["ci_wiggle"])
        assert (got["ciposlow"], got["ciposhigh"], got["ciendlow"], got["ciendhigh"]) == (-w, w, -w, w), label
        assert (got["insLen"], got["consBp"], got["homLen"]) == (int(rr["ins_len"]), int(rr["cons_bp"]), int(rr["hom_len"])), label
        assert np.float32(got["q"]) == rr["sr_align_quality"], label
        assert got["precise"] == 1 and got["alleles"] == pyoracle.blob_field(rr, rb, "allele"), label
    else:
        assert (got["svStart"], got["svEnd"], got["precise"]) == (int(J["sv_start"]), int(J["sv_end"]), 0), label


def test_cpp_msa_align_consensus_and_refine_batch_vs_reference(tmp_path, reference):
    b = synth.make_batch(48, mode="mixed", n_reads=6, seed=17)
    P = abi.params_sr()
    rd = _run(tmp_path, b, P, 0, 1)
    rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=8)
    assert rd.take("i") == b.n
    n_ok = 0
    for k in range(b.n):
        got = rd.sv()
        if int(b.junctions["n_seq"][k]) <= 1:
            assert got["ok"] == 0
            continue
        assert got["rows"] == int(rr["sr_support"][k])
        _check_sv(got, b.junctions[k], rr[k], rb, "per-call %d" % k)
        n_ok += got["ok"]
    assert n_ok > 0.7 * b.n
    nb = rd.take("i")
    assert nb == int((b.junctions["n_seq"] > 1).sum())
    for _ in range(nb):
        k = rd.take("i")
        got = rd.sv()
        assert got["ok"] == int(rr["ok"][k])
        if got["ok"]:
            _check_sv(got, b.junctions[k], rr[k], rb, "refineBatch %d" % k)
            assert got["srSupport"] == int(b.junctions["n_seq"][k])     # src/shortpe.h:196
        else:
            assert got["consensus"] == b"" and got["srSupport"] == 0 and got["q"] == 0.0   # src/shortpe.h:186-190
    # the primitives
    f1 = rd.take("i")
    cons, ref, r0, r1 = rd.s(), rd.s(), rd.s(), rd.s()
    want = reference.long_needle(cons, ref)
    assert (bool(f1), r0, r1) == (bool(want[0]), want[1], want[2]) and f1 == 1
    f2 = rd.take("i")
    icons, iref, s0, s1 = rd.s(), rd.s(), rd.s(), rd.s()
    rc, c_row, r_row, _ = reference.split_align(icons, iref)
    assert (f2, rc) == (1, 1)
    assert (s0, s1) == (r_row, c_row)      # splitAlign's own orientation: reference row first (src/split.h:546-552 swaps)
    sc, nrows = rd.take("ii")
    rows = [rd.s() for _ in range(nrows)]
    up = ref
    wsc, wrows = reference.gotoh([up[500:560]], [up[495:565]])
    assert (sc, rows) == (wsc, wrows)
    a, bb, d = rd.s(), rd.s(), rd.s()
    src = b"ACGTNacgtRYKM"
    assert a == reference.reverse_complement(src) and bb == src and d == reference.reverse_complement(src)
    # std::unordered_set<std::string> as the read container -- duplicates collapse, iteration order is the input order
    nset = rd.take("i")
    assert nset == 6
    order_matters = 0
    for k in range(nset):
        reads = b.seqs_of(k)
        size, rows = rd.take("ii")
        cs = rd.s()
        perm = reference.unordered_set_order(reads + [reads[0]])
        assert size == len(perm) == len(set(reads))
        ordered = [(reads + [reads[0]])[i] for i in perm]
        want_rows, want_cs = reference.msa(ordered)
        assert (rows, cs) == (want_rows, want_cs), k
        order_matters += reference.msa(reads)[1] != want_cs
    # (the consensus usually does not depend on the order of six clean reads; the contract is that the device sees the set's order)


def test_cpp_long_read_entry_points_vs_reference(tmp_path, reference):
    P = abi.params_lr(realign=True)
    for kw in (dict(mode="lr", n_reads=5, sub_rate=0.05), dict(mode="lrins", n_reads=6, sub_rate=0.04)):
        b = synth.make_batch(4, seed=29, **kw)
        rd = _run(tmp_path, b, P, 1, 2)
        rr, rb = reference.refine_batch(b, params=P, want_alignment=False, n_threads=4)
        assert rd.take("i") == b.n
        for k in range(b.n):
            got = rd.sv()
            assert got["rows"] == int(rr["sr_support"][k]), kw
            _check_sv(got, b.junctions[k], rr[k], rb, "%s %d" % (kw["mode"], k))
        assert int(rr["ok"].sum()) >= 3


def test_cpp_align_consensus_only_all_types_vs_reference(tmp_path, reference):
    b = synth.make_batch(60, mode="mixed", seed=5)
    rd = _run(tmp_path, b, abi.params_sr(), 0, 0)
    rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=8)
    assert rd.take("i") == b.n
    for k in range(b.n):
        _check_sv(rd.sv(), b.junctions[k], rr[k], rb, "U %d" % k, with_cons=False)
