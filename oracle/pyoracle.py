"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of the two checkers:

  Oracle("port")      oracle/liboracle.so          plain-C restatement
  Oracle("reference") oracle/_ref/libdelly_ref.so  the reference's own headers

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (delly_amd/, libdellyhip.so) never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from delly_amd import abi  # noqa: E402  (shared record layouts only)

PORT_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdelly_ref.so")


def build(force=False):
    """Compiles the C restatement, and oracle/_ref when /root/reference exists."""
    if force or not os.path.exists(PORT_SO) or \
            os.path.getmtime(PORT_SO) < os.path.getmtime(os.path.join(HERE, "delly_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def have_reference():
    return os.path.exists(REF_SO)


def _u8(a):
    return np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                                dtype=np.uint8)


def _p(a, typ=C.c_char_p):
    return a.ctypes.data_as(typ)


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            if not os.path.exists(PORT_SO):
                build()
            self.lib = C.CDLL(PORT_SO)
            self.pre = "dor_"
        elif kind == "reference":
            self.lib = C.CDLL(REF_SO)
            self.pre = "dref_"
        else:
            raise ValueError(kind)
        self.params = abi.params_sr()

    def _f(self, name):
        return getattr(self.lib, self.pre + name)

    # --- primitives ---------------------------------------------------------
    def lcs(self, a, b):
        a, b = _u8(a), _u8(b)
        return self._f("lcs")(_p(a), a.size, _p(b), b.size)

    def reverse_complement(self, s):
        a = _u8(s).copy()
        f = self._f("reverse_complement")
        f.restype = None
        f(_p(a), a.size)
        return a.tobytes()

    def longest_homology(self, a, b, thr=-1):
        a, b = _u8(a), _u8(b)
        return self._f("longest_homology")(_p(a), a.size, _p(b), b.size, thr)

    def long_needle(self, s1, s2):
        """-> (found, row0, row1, diag or None)"""
        s1, s2 = _u8(s1), _u8(s2)
        cap = s1.size + s2.size + 8
        rows = np.zeros(2 * cap, dtype=np.uint8)
        ln = C.c_int(0)
        if self.kind == "port":
            diag = (C.c_int * 5)()
            rc = self._f("long_needle")(_p(s1), s1.size, _p(s2), s2.size, _p(rows), cap, C.byref(ln), diag)
            d = list(diag)
        else:
            rc = self._f("long_needle")(_p(s1), s1.size, _p(s2), s2.size, _p(rows), cap, C.byref(ln))
            d = None
        assert rc >= 0
        L = ln.value
        return bool(rc), rows[:L].tobytes(), rows[cap:cap + L].tobytes(), d

    def edlib_align(self, q, t, mode, task=2):
        """edlibAlign(q, t, {k=-1, mode, task}); mode 0 NW / 1 SHW / 2 HW; task 0 DISTANCE / 1 LOC / 2 PATH
        -> (editDistance, numLocations, endLocations[0], startLocations[0], ops bytes) or None (port: Hirschberg regime)"""
        q, t = _u8(q), _u8(t)
        cap = q.size + t.size + 8
        aln = np.zeros(cap, dtype=np.uint8)
        out = (C.c_int * 4)()
        L = self._f("edlib_align")(_p(q), q.size, _p(t), t.size, mode, task, out, _p(aln, C.POINTER(C.c_ubyte)), cap)
        if L < 0:
            return None
        o = list(out)
        if self.kind == "port":  # the port reports numLocations only for the modes that enumerate them
            pass
        return o[0], o[1], o[2], o[3], aln[:L].tobytes()

    def edlib_align_full(self, q, t, k=-1, mode=0, task=2, iupac=False):
        """the reference's edlibAlign with every field of its result (reference only) ->
        dict(status, ed, ends, starts (None when absent), ops, alphabet)"""
        q, t = _u8(q), _u8(t)
        cap = t.size + 2
        acap = q.size + t.size + 8
        aln = np.zeros(acap, dtype=np.uint8)
        ends, starts = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
        out = (C.c_int * 5)()
        self._f("edlib_align_full")(_p(q), q.size, _p(t), t.size, int(k), int(mode), int(task), int(bool(iupac)), out,
                                    _p(ends, C.POINTER(C.c_int)), _p(starts, C.POINTER(C.c_int)), cap, _p(aln, C.POINTER(C.c_ubyte)), acap)
        n = out[2]
        return dict(status=out[0], ed=out[1], ends=ends[:n].tolist(), starts=(None if (n == 0 or starts[0] == -2) else starts[:n].tolist()),
                    ops=aln[:out[3]].tobytes(), alphabet=out[4])

    def edlib_cigar(self, ops, fmt):
        a = _u8(ops)
        buf = C.create_string_buffer(4 * a.size + 16)
        n = self._f("edlib_cigar")(_p(a, C.POINTER(C.c_ubyte)), a.size, int(fmt), buf, len(buf))
        return None if n < 0 else buf.value

    def split_align(self, cons, ref):
        """splitAlign + row swap -> (rc, cons_row, ref_row, internals or None)"""
        cons, ref = _u8(cons), _u8(ref)
        cap = cons.size + ref.size + 16
        rows = np.zeros(2 * cap, dtype=np.uint8)
        ln = C.c_int(0)
        if self.kind == "port":
            internals = (C.c_int * 5)()
            rc = self._f("split_align")(_p(cons), cons.size, _p(ref), ref.size, _p(rows), cap, C.byref(ln), internals)
            d = list(internals)
        else:
            rc = self._f("split_align")(_p(cons), cons.size, _p(ref), ref.size, _p(rows), cap, C.byref(ln))
            d = None
        L = ln.value
        return rc, rows[:L].tobytes(), rows[cap:cap + L].tobytes(), d

    def gotoh(self, a1, a2):
        """a1, a2: lists of equal-length byte strings (alignment rows) -> (score, rows)"""
        r1, m = len(a1), len(a1[0])
        r2, n = len(a2), len(a2[0])
        A1 = _u8(b"".join(a1))
        A2 = _u8(b"".join(a2))
        cap = m + n + 8
        out = np.zeros((r1 + r2) * cap, dtype=np.uint8)
        ln = C.c_int(0)
        score = self._f("gotoh")(C.byref(self.params), _p(A1), r1, m, _p(A2), r2, n, _p(out), cap, C.byref(ln))
        L = ln.value
        return score, [out[i * cap:i * cap + L].tobytes() for i in range(r1 + r2)]

    def consensus(self, rows):
        r, m = len(rows), len(rows[0])
        A = _u8(b"".join(rows))
        cs = np.zeros(m + 1, dtype=np.uint8)
        L = self._f("consensus")(C.byref(self.params), _p(A), r, m, _p(cs), m + 1)
        return cs[:L].tobytes()

    @staticmethod
    def _pack(reads):
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        return _u8(b"".join(reads)), off

    def guide_tree(self, reads):
        blob, off = self._pack(reads)
        n = len(reads)
        D = 2 * n + 1
        d = np.zeros(D * D, dtype=np.int32)
        p = np.zeros(D * 3, dtype=np.int32)
        root = self._f("guide_tree")(n, _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(d, C.POINTER(C.c_int)),
                                     _p(p, C.POINTER(C.c_int)))
        return root, d.reshape(D, D), p.reshape(D, 3)

    def msa(self, reads):
        """-> (rows, consensus)"""
        blob, off = self._pack(reads)
        cap = int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln = C.c_int(0)
        rows = self._f("msa")(C.byref(self.params), len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(cs),
                              cap, C.byref(ln))
        return rows, cs[:ln.value].tobytes()

    def msa_edlib(self, reads):
        """msaEdlib(c, sps, cs) -> (rows, consensus)"""
        blob, off = self._pack(reads)
        cap = int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln = C.c_int(0)
        rows = self._f("msa_edlib")(C.byref(self.params), len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(cs),
                                    cap, C.byref(ln))
        return rows, cs[:ln.value].tobytes()

    def msa_wfa(self, reads, prefix=b"", suffix=b""):
        """msaWfa(c, sps, cs, prefix, suffix) -> (rows, consensus)"""
        blob, off = self._pack(reads)
        cap = 2 * int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln = C.c_int(0)
        pre, suf = _u8(prefix), _u8(suffix)
        rows = self._f("msa_wfa")(C.byref(self.params), len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)),
                                  _p(pre), pre.size, _p(suf), suf.size, _p(cs), cap, C.byref(ln))
        return rows, cs[:ln.value].tobytes()

    def classify_reads(self, jobs, blob, params=None, n_threads=1, with_dist=True):
        """worker body of process_batch (src/coverage.h:418-434) for every job -> structured result array"""
        p = params if params is not None else self.params
        blob = _u8(blob)
        jobs = np.ascontiguousarray(jobs, dtype=abi.align_job_dtype())
        res = np.zeros(jobs.shape[0], dtype=abi.align_result_dtype())
        sec = C.c_double(0)
        rc = self._f("classify_reads")(C.byref(p), C.c_uint64(jobs.shape[0]), C.c_void_p(jobs.ctypes.data), _p(blob),
                                       C.c_void_p(res.ctypes.data), int(n_threads), int(bool(with_dist)), C.byref(sec))
        self.worker_seconds = sec.value   # the parallel region alone (bench.py cpu_baseline leg)
        if rc:
            raise RuntimeError("oracle classify_reads rc=%d" % rc)
        return res

    def edit_distance_nw_batch(self, jobs, blob, n_threads=1):
        """_editDistanceNW (src/genotype.h:21-30) for every pair -> int32 distances"""
        blob = _u8(blob)
        jobs = np.ascontiguousarray(jobs, dtype=abi.nw_job_dtype())
        out = np.zeros(jobs.shape[0], dtype=np.int32)
        sec = C.c_double(0)
        rc = self._f("edit_distance_nw_batch")(C.c_uint64(jobs.shape[0]), C.c_void_p(jobs.ctypes.data), _p(blob),
                                               C.c_void_p(out.ctypes.data), int(n_threads), C.byref(sec))
        self.worker_seconds = sec.value
        if rc:
            raise RuntimeError("oracle edit_distance_nw_batch rc=%d" % rc)
        return out

    def generate_probes(self, batch, params=None):
        """per-SV body of _generateProbes (src/coverage.h:196-258) over a synth.Batch with given consensus
        -> (probes structured array, blob np.uint8)"""
        p = params if params is not None else self.params
        n = batch.n
        nchr = len(batch.chroms)
        chr_ptrs = (C.c_char_p * nchr)(*[C.cast(c.ctypes.data, C.c_char_p) for c in batch.chroms])
        chr_len = np.array([c.size for c in batch.chroms], dtype=np.int64)
        rec = np.zeros(n, dtype=abi.probes_dtype())
        cap = n * 4096 + 64
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        junc = np.ascontiguousarray(batch.junctions)
        blob = _u8(batch.seq_blob)
        off = np.ascontiguousarray(batch.seq_off, dtype=np.uint64)
        rc = self._f("generate_probes")(C.byref(p), nchr, chr_ptrs, _p(chr_len, C.POINTER(C.c_int64)), n,
                                        C.c_void_p(junc.ctypes.data), _p(blob), _p(off, C.POINTER(C.c_uint64)),
                                        C.c_void_p(rec.ctypes.data), _p(out), C.c_uint64(cap), C.byref(used))
        if rc:
            raise RuntimeError("oracle generate_probes rc=%d" % rc)
        return rec, out[:used.value]

    def unordered_set_order(self, reads):
        assert self.kind == "reference"
        blob, off = self._pack(reads)
        perm = np.zeros(len(reads), dtype=np.int32)
        k = self.lib.dref_unordered_set_order(len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)),
                                              _p(perm, C.POINTER(C.c_int)))
        return [int(x) for x in perm[:k]]

    # --- batch --------------------------------------------------------------
    def refine_batch(self, batch, with_msa=None, want_alignment=True, n_threads=1, params=None):
        """Runs msa+alignConsensus (or alignConsensus only) over a synth.Batch.
        -> (results structured array, out_blob np.uint8)"""
        if with_msa is None:
            with_msa = batch.with_msa
        p = params if params is not None else self.params
        n = batch.n
        nchr = len(batch.chroms)
        chr_ptrs = (C.c_char_p * nchr)(*[C.cast(c.ctypes.data, C.c_char_p) for c in batch.chroms])
        chr_len = np.array([c.size for c in batch.chroms], dtype=np.int64)
        res = np.zeros(n, dtype=abi.result_dtype())
        cap = int(batch.seq_blob.size) * 3 + n * 16384 + (int(batch.seq_blob.size) * 16 + n * 65536 if want_alignment else 0)
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        junc = np.ascontiguousarray(batch.junctions)
        rc = self._f("refine_batch")(
            C.byref(p), nchr, chr_ptrs, _p(chr_len, C.POINTER(C.c_int64)), n,
            _p(junc, C.c_void_p), _p(batch.seq_blob), _p(batch.seq_off, C.POINTER(C.c_uint64)),
            _p(res, C.c_void_p), _p(out), C.c_uint64(cap), C.byref(used), int(with_msa),
            int(bool(want_alignment)), int(n_threads))
        assert rc == 0, "oracle out_blob overflow"
        return res, out[:used.value]


def _time_refine(self, batch, with_msa=None, n_threads=1, reps=1, params=None):
    """bench.py cpu_baseline: `reps` passes of the loop body of src/shortpe.h:183-197 over the batch on n_threads
    std::threads (atomic work counter), timed INSIDE the C/C++ function around thread start .. join; no diagnostic
    replay, no result marshalling.  -> (seconds, junction visits, alignConsensus()==true count)"""
    if with_msa is None:
        with_msa = batch.with_msa
    p = params if params is not None else self.params
    nchr = len(batch.chroms)
    chr_ptrs = (C.c_char_p * nchr)(*[C.cast(c.ctypes.data, C.c_char_p) for c in batch.chroms])
    chr_len = np.array([c.size for c in batch.chroms], dtype=np.int64)
    junc = np.ascontiguousarray(batch.junctions)
    sec = C.c_double(0)
    oks = C.c_int64(0)
    rc = self._f("time_refine_batch")(
        C.byref(p), nchr, chr_ptrs, _p(chr_len, C.POINTER(C.c_int64)), batch.n, _p(junc, C.c_void_p),
        _p(batch.seq_blob), _p(batch.seq_off, C.POINTER(C.c_uint64)), int(with_msa), int(n_threads), int(reps),
        C.byref(sec), C.byref(oks))
    assert rc == 0
    return sec.value, int(reps) * batch.n, oks.value


Oracle.time_refine = _time_refine


def blob_field(res_row, blob, which):
    """Extracts 'cons' | 'allele' | 'aln' bytes of one result row."""
    if which == "cons":
        o, l = int(res_row["cons_off"]), int(res_row["cons_len"])
    elif which == "allele":
        o, l = int(res_row["allele_off"]), int(res_row["allele_len"])
    else:
        o, l = int(res_row["aln_off"]), 2 * int(res_row["aln_len"])
    return blob[o:o + l].tobytes() if l > 0 else b""
