"""Host-side mirror of the reference's interface for the split-read refinement
path, on top of the C-ABI in include/dellyhip.h (libdellyhip.so, hand-written
HIP for gfx950).  Names follow the reference:

    msa(c, sps, cs)                      src/msa.h:185-239
    alignConsensus(c, hdr, seq, ...)     src/split.h:644-672
    longNeedle(s1, s2, align, ...)       src/needle.h:45-222
    gotoh(a1, a2, align, ...)            src/gotoh.h:71-174
    lcs(s1, s2)                          src/msa.h:10-30

There is no CPU path here: if the shared library or a gfx950 device is missing
every call raises DellyHipError.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdellyhip.so")

EXPORTS = [
    "dellyhip_recut_alleles", "dellyhip_recut_alleles_batch", "dellyhip_stream_zero_copy",
    "dellyhip_create", "dellyhip_destroy", "dellyhip_last_error", "dellyhip_default_params_sr",
    "dellyhip_default_params_lr", "dellyhip_set_chromosome", "dellyhip_refine_batch",
    "dellyhip_align_consensus_batch", "dellyhip_batch_upload", "dellyhip_batch_run", "dellyhip_batch_sync",
    "dellyhip_batch_fetch", "dellyhip_batch_fetch_begin", "dellyhip_batch_fetch_end", "dellyhip_batch_free", "dellyhip_batch_kernel_ms", "dellyhip_batch_device_results", "dellyhip_batch_dp_kernel_ms", "dellyhip_long_needle",
    "dellyhip_lcs", "dellyhip_gotoh", "dellyhip_msa", "dellyhip_abi_info", "dellyhip_edlib_align", "dellyhip_refine_batch_lr", "dellyhip_msa_edlib", "dellyhip_msa_wfa",
    "dellyhip_classify_reads", "dellyhip_jobs_upload", "dellyhip_jobs_run", "dellyhip_jobs_sync", "dellyhip_jobs_fetch",
    "dellyhip_jobs_free", "dellyhip_jobs_kernel_ms",
    "dellyhip_edit_distance_nw_batch", "dellyhip_nwjobs_upload", "dellyhip_nwjobs_run", "dellyhip_nwjobs_fetch",
    "dellyhip_nwjobs_free", "dellyhip_nwjobs_kernel_ms", "dellyhip_generate_probes_batch", "dellyhip_batch_probes",
    "dellyhip_split_align", "dellyhip_shard_by_cost", "dellyhip_comm_unique_id", "dellyhip_comm_create", "dellyhip_comm_destroy",
    "dellyhip_gather_results", "dellyhip_gather_results_device",
    "dellyhip_create_shared", "dellyhip_trim_memory", "dellyhip_compute_streams", "dellyhip_host_register", "dellyhip_host_unregister", "dellyhip_stream_create", "dellyhip_stream_destroy", "dellyhip_stream_submit", "dellyhip_stream_collect",
    "dellyhip_stream_pending", "dellyhip_stream_release", "dellyhip_stream_stats", "dellyhip_batch_sparse_left", "dellyhip_batch_lr_team_stats", "dellyhip_batch_msa_stats", "dellyhip_rebase_gathered",
    "dellyhip_comm_create_hostlink", "dellyhip_comm_info", "dellyhip_comm_exchange_sizes", "dellyhip_comm_exchange_ready", "dellyhip_comm_gather_bytes", "dellyhip_edlib_align_full",
]


class DellyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dellyhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def recut_alleles(params, junctions, results, blob, chroms):
    """dellyhip_recut_alleles over a batch: the "REF,ALT" bytes of every record of a compact-payload run (allele_len < 0), cut on the
    host from the chromosomes and the records' consensus bytes exactly as src/split.h:606-624 -> list of bytes (b"" where the record
    has no compact alleles).  junctions: the batch AS SUBMITTED."""
    buf, off, _ = recut_alleles_raw(params, junctions, results, blob, chroms)
    raw = buf.tobytes()
    return [raw[int(off[i]):int(off[i + 1])] for i in range(results.shape[0])]


def recut_alleles_raw(params, junctions, results, blob, chroms, out=None):
    """dellyhip_recut_alleles_batch -> (bytes as a uint8 array, offsets (n + 1), seconds inside the C function)"""
    import time
    lib = load_library()
    lib.dellyhip_recut_alleles_batch.restype = C.c_int64
    n = int(results.shape[0])
    junctions = np.ascontiguousarray(junctions)
    results = np.ascontiguousarray(results)
    need = int(np.maximum(-results["allele_len"].astype(np.int64), 0).sum())
    if out is None or out.nbytes < need:
        out = np.empty(max(need, 1), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    ptrs = (C.c_void_p * len(chroms))(*[c.ctypes.data for c in chroms])
    lens = np.array([c.size for c in chroms], dtype=np.int64)
    t0 = time.perf_counter()
    got = lib.dellyhip_recut_alleles_batch(C.byref(params), n, C.c_void_p(junctions.ctypes.data), C.c_void_p(results.ctypes.data),
                                           C.c_void_p(blob.ctypes.data), ptrs, C.c_void_p(lens.ctypes.data), len(chroms),
                                           C.c_void_p(out.ctypes.data), C.c_uint64(out.nbytes), C.c_void_p(off.ctypes.data))
    dt = time.perf_counter() - t0
    if got != need:
        raise DellyHipError(int(got) if got < 0 else abi.E_ARG, "dellyhip_recut_alleles_batch: " + (lib.dellyhip_last_error() or b"").decode())
    return out[:need], off, dt


def load_library():
    global _lib
    if _lib is None:
        path = os.environ.get("DELLYHIP_LIB", LIB_PATH)   # (override: A/B runs of tuning builds)
        if not os.path.exists(path):
            raise DellyHipError(abi.E_NODEVICE, "libdellyhip.so is not built (run __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.dellyhip_last_error.restype = C.c_char_p
        lib.dellyhip_batch_free.restype = None
        lib.dellyhip_destroy.restype = None
        lib.dellyhip_jobs_free.restype = None
        lib.dellyhip_nwjobs_free.restype = None
        lib.dellyhip_comm_destroy.restype = None
        lib.dellyhip_stream_destroy.restype = None
        lib.dellyhip_stream_release.restype = None
        lib.dellyhip_stream_stats.restype = None
        lib.dellyhip_trim_memory.restype = C.c_uint64
        lib.dellyhip_trim_memory.argtypes = [C.c_void_p]
        # explicit prototypes wherever a 64-bit integer or a pointer could otherwise travel as a default C int
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
        lib.dellyhip_stream_create.argtypes = [vp, i32, i32, i32, vp]
        lib.dellyhip_stream_destroy.argtypes = [vp]
        lib.dellyhip_stream_submit.argtypes = [vp, i32, vp, vp, vp, u64, u64]
        lib.dellyhip_stream_collect.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.dellyhip_stream_pending.argtypes = [vp]
        lib.dellyhip_stream_release.argtypes = [vp]
        lib.dellyhip_host_register.argtypes = [vp, vp, u64]
        lib.dellyhip_host_unregister.argtypes = [vp, vp]
        lib.dellyhip_comm_create_hostlink.argtypes = [vp, C.c_char_p, i32, i32, vp]
        lib.dellyhip_comm_info.argtypes = [vp, vp, vp, vp, vp]
        lib.dellyhip_comm_exchange_sizes.argtypes = [vp, vp, u64, u64, i32, vp]
        lib.dellyhip_comm_exchange_ready.argtypes = [vp, vp, i32, i32]
        lib.dellyhip_comm_gather_bytes.argtypes = [vp, vp, i32, vp, u64, vp, u64, vp]
        _lib = lib
    return _lib


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a, typ=C.c_char_p):
    return a.ctypes.data_as(typ)


def shard_by_cost(junctions, seq_off, world, params=None):
    """owner[i] = rank of junction i, balanced by predicted cost (dellyhip_shard_by_cost; pure host arithmetic, no GPU)."""
    lib = load_library()
    p = params if params is not None else abi.params_sr()
    junc = np.ascontiguousarray(junctions)
    off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    owner = np.zeros(junc.shape[0], dtype=np.int32)
    rc = lib.dellyhip_shard_by_cost(C.byref(p), int(junc.shape[0]), _p(junc, C.c_void_p), _p(off, C.POINTER(C.c_uint64)),
                                    C.c_uint64(off.size - 1), int(world), _p(owner, C.POINTER(C.c_int32)))
    if rc != 0:
        raise DellyHipError(rc, lib.dellyhip_last_error().decode())
    return owner


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it and hands it to the other ranks)."""
    lib = load_library()
    buf = (C.c_ubyte * 128)()
    rc = lib.dellyhip_comm_unique_id(buf)
    if rc != 0:
        raise DellyHipError(rc, lib.dellyhip_last_error().decode())
    return bytes(buf)


class Comm:
    """One communicator per process (dellyhip_comm_create: RCCL, one process per GPU; world == 1 needs no id).
    hostlink=<name>: the same communicator over POSIX shared memory (dellyhip_comm_create_hostlink) -- the ranks of one
    node, also several per GPU; ctx=None gives a device-less one for the protocol exchanges only."""

    def __init__(self, ctx, rank=0, world=1, unique_id=None, hostlink=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.lib = ctx.lib if ctx is not None else load_library()
        self._c = C.c_void_p()
        cx = ctx._ctx if ctx is not None else None
        if hostlink is not None:
            rc = self.lib.dellyhip_comm_create_hostlink(cx, str(hostlink).encode(), self.rank, self.world, C.byref(self._c))
        else:
            idbuf = (C.c_ubyte * 128).from_buffer_copy(unique_id) if unique_id is not None else None
            rc = self.lib.dellyhip_comm_create(cx, idbuf, self.rank, self.world, C.byref(self._c))
        self._check(rc)

    def _check(self, rc):
        if rc != 0:
            raise DellyHipError(rc, self.lib.dellyhip_last_error().decode())

    def info(self):
        """-> dict(rank, world, transport_ranks = what RCCL / the hostlink itself reports, kind)"""
        r, w, t = C.c_int32(), C.c_int32(), C.c_int32()
        kind = C.create_string_buffer(16)
        self._check(self.lib.dellyhip_comm_info(self._c, C.byref(r), C.byref(w), C.byref(t), kind))
        return {"rank": r.value, "world": w.value, "transport_ranks": t.value, "kind": kind.value.decode()}

    def exchange_sizes(self, count, nbytes, failed=False):
        """first collective step of the gather (dellyhip_comm_exchange_sizes) -> [(count, bytes)] of every rank; raises on
        EVERY rank if any rank reports a failure"""
        out = (C.c_uint64 * (2 * self.world))()
        cx = self.ctx._ctx if self.ctx is not None else None
        self._check(self.lib.dellyhip_comm_exchange_sizes(cx, self._c, C.c_uint64(int(count)), C.c_uint64(int(nbytes)), int(bool(failed)), out))
        return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(self.world)]

    def exchange_ready(self, root=0, root_failed=False):
        """second collective step (dellyhip_comm_exchange_ready): raises on every rank if the root could not size its buffers"""
        cx = self.ctx._ctx if self.ctx is not None else None
        self._check(self.lib.dellyhip_comm_exchange_ready(cx, self._c, int(root), int(bool(root_failed))))

    def gather_bytes(self, payload, root=0, cap=None):
        """dellyhip_comm_gather_bytes on HOST memory (device-less hostlink): payload = bytes-like -> (bytes on the root / None
        elsewhere, sizes of every rank)"""
        buf = np.frombuffer(bytes(payload), dtype=np.uint8)
        cap = int(cap if cap is not None else 1 << 24)
        out = np.zeros(cap if self.rank == root else 0, dtype=np.uint8)
        sizes = (C.c_uint64 * self.world)()
        cx = self.ctx._ctx if self.ctx is not None else None
        self._check(self.lib.dellyhip_comm_gather_bytes(cx, self._c, int(root), buf.ctypes.data if buf.size else None, C.c_uint64(buf.size),
                                                        out.ctypes.data if out.size else None, C.c_uint64(out.size), sizes))
        sz = [int(x) for x in sizes]
        return (out[:sum(sz)].tobytes() if self.rank == root else None), sz

    def close(self):
        if self._c:
            self.lib.dellyhip_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One context per GPU (per process rank): replaces the ThreadPool of src/shortpe.h:80."""

    def __init__(self, params=None, device=0, share_with=None):
        """share_with: another Context whose resident chromosomes this one shares (dellyhip_create_shared)"""
        self.lib = load_library()
        self._ctx = C.c_void_p()
        if share_with is not None:
            self.params = params if params is not None else share_with.params
            rc = self.lib.dellyhip_create_shared(share_with._ctx, C.byref(self.params), C.byref(self._ctx))
        else:
            self.params = params if params is not None else abi.params_sr()
            rc = self.lib.dellyhip_create(C.byref(self.params), int(device), C.byref(self._ctx))
        self._check(rc)
        self._chroms = []

    def _check(self, rc):
        if rc != 0:
            raise DellyHipError(rc, self.lib.dellyhip_last_error().decode())

    def close(self):
        if self._ctx:
            self.lib.dellyhip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def compute_streams(self):
        """two hipStream_t handles (ints) of this device verified to run side by side (dellyhip_compute_streams)"""
        out = (C.c_void_p * 2)()
        self._check(self.lib.dellyhip_compute_streams(self._ctx, out))
        return int(out[0] or 0), int(out[1] or 0)

    def host_register(self, address, nbytes):
        """pins caller memory (e.g. a shared-memory segment) for fetch / gather destinations (dellyhip_host_register)"""
        self._check(self.lib.dellyhip_host_register(self._ctx, C.c_void_p(address), C.c_uint64(nbytes)))

    def host_unregister(self, address):
        self._check(self.lib.dellyhip_host_unregister(self._ctx, C.c_void_p(address)))

    def trim_memory(self):
        """returns the device / pinned blocks the library keeps parked to the HIP runtime (dellyhip_trim_memory) -> bytes"""
        return int(self.lib.dellyhip_trim_memory(self._ctx))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # hdr->target_len[chr] + faidx_fetch_seq buffer (src/shortpe.h:88)
    def set_chromosome(self, chr_index, seq):
        seq = _u8(seq)
        self._check(self.lib.dellyhip_set_chromosome(self._ctx, int(chr_index), _p(seq), C.c_int64(seq.size)))

    def set_chromosomes(self, chroms):
        for i, s in enumerate(chroms):
            self.set_chromosome(i, s)

    # ---- batched hot path ---------------------------------------------------
    def _run_host(self, fn, junctions, seq_blob, seq_off, want_alignment):
        n = int(junctions.shape[0])
        junc = np.ascontiguousarray(junctions)
        blob = _u8(seq_blob)
        off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        res = np.zeros(n, dtype=abi.result_dtype())
        # consensus + "REF,ALT" (+ two alignment rows): bounded by a few times the sequence bytes per junction
        cap = n * (6000 if want_alignment else 3100) + int(blob.size) * (16 if want_alignment else 8) + 64
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        rc = fn(self._ctx, n, _p(junc, C.c_void_p), _p(blob), _p(off, C.POINTER(C.c_uint64)),
                C.c_uint64(off.size - 1), _p(res, C.c_void_p), _p(out), C.c_uint64(cap), C.byref(used),
                int(bool(want_alignment)))
        self._check(rc)
        return res, out[:used.value]

    def align_consensus_batch(self, junctions, seq_blob, seq_off, want_alignment=False):
        """alignConsensus(c, hdr, seq, sndSeq, sv) for every junction (unit U)."""
        return self._run_host(self.lib.dellyhip_align_consensus_batch, junctions, seq_blob, seq_off, want_alignment)

    def refine_batch(self, junctions, seq_blob, seq_off, want_alignment=False):
        """msa() + alignConsensus() for every junction: loop body of src/shortpe.h:183-197."""
        return self._run_host(self.lib.dellyhip_refine_batch, junctions, seq_blob, seq_off, want_alignment)

    def refine_batch_lr(self, junctions, seq_blob, seq_off, want_alignment=False):
        """msaEdlib() + alignConsensus(..., realign): loop body of src/assemble.h:833-872 (non-insertion junctions)."""
        return self._run_host(self.lib.dellyhip_refine_batch_lr, junctions, seq_blob, seq_off, want_alignment)

    def classify_reads(self, jobs, blob):
        """The worker body of process_batch (src/coverage.h:418-434) for every AlignJob -> AlignResult records."""
        jobs = np.ascontiguousarray(jobs, dtype=abi.align_job_dtype())
        blob = _u8(blob)
        res = np.zeros(jobs.shape[0], dtype=abi.align_result_dtype())
        self._check(self.lib.dellyhip_classify_reads(self._ctx, C.c_uint64(jobs.shape[0]), _p(jobs, C.c_void_p), _p(blob),
                                                     C.c_uint64(blob.size), _p(res, C.c_void_p)))
        return res

    def edit_distance_nw_batch(self, jobs, blob):
        """_editDistanceNW (src/genotype.h:21-30) for every (query, target) pair -> int32 distances."""
        jobs = np.ascontiguousarray(jobs, dtype=abi.nw_job_dtype())
        blob = _u8(blob)
        out = np.zeros(jobs.shape[0], dtype=np.int32)
        self._check(self.lib.dellyhip_edit_distance_nw_batch(self._ctx, C.c_uint64(jobs.shape[0]), _p(jobs, C.c_void_p),
                                                             _p(blob), C.c_uint64(blob.size), _p(out, C.c_void_p)))
        return out

    def generate_probes(self, batch):
        """The per-SV body of _generateProbes (src/coverage.h:196-258) for a synth.Batch with the consensus given
        -> (probes structured array, blob np.uint8)"""
        n = batch.n
        junc = np.ascontiguousarray(batch.junctions)
        blob = _u8(batch.seq_blob)
        off = np.ascontiguousarray(batch.seq_off, dtype=np.uint64)
        rec = np.zeros(n, dtype=abi.probes_dtype())
        cap = n * 4 * 640 + 64
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        self._check(self.lib.dellyhip_generate_probes_batch(self._ctx, n, _p(junc, C.c_void_p), _p(blob),
                                                            _p(off, C.POINTER(C.c_uint64)), C.c_uint64(off.size - 1),
                                                            _p(rec, C.c_void_p), _p(out), C.c_uint64(cap), C.byref(used)))
        return rec, out[:used.value]

    def refine(self, batch, want_alignment=False):
        """Convenience for a synth.Batch."""
        if batch.with_msa == 2:
            return self.refine_batch_lr(batch.junctions, batch.seq_blob, batch.seq_off, want_alignment)
        fn = self.refine_batch if batch.with_msa else self.align_consensus_batch
        return fn(batch.junctions, batch.seq_blob, batch.seq_off, want_alignment)

    # ---- device-resident batches --------------------------------------------
    def upload(self, batch):
        return ResidentBatch(self, batch)

    # ---- single-item wrappers -------------------------------------------------
    def long_needle(self, s1, s2):
        """-> (found, row0, row1)"""
        s1, s2 = _u8(s1), _u8(s2)
        cap = s1.size + s2.size + 8
        rows = np.zeros(2 * cap, dtype=np.uint8)
        ln, found = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.dellyhip_long_needle(self._ctx, _p(s1), s1.size, _p(s2), s2.size, _p(rows), cap,
                                                  C.byref(ln), C.byref(found)))
        L = ln.value if found.value else 0
        return bool(found.value), rows[:L].tobytes(), rows[cap:cap + L].tobytes()

    def split_align(self, cons, ref):
        """splitAlign + the row swap of _consRefAlignment (src/split.h:480-552) -> (found, cons row, ref row)"""
        s1, s2 = _u8(cons), _u8(ref)
        cap = s1.size + s2.size + 8
        rows = np.zeros(2 * cap, dtype=np.uint8)
        ln, found = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.dellyhip_split_align(self._ctx, _p(s1), s1.size, _p(s2), s2.size, _p(rows), cap,
                                                  C.byref(ln), C.byref(found)))
        L = ln.value if found.value else 0
        return bool(found.value), rows[:L].tobytes(), rows[cap:cap + L].tobytes()

    def edlib_align(self, q, t, mode, task=2):
        """edlibAlign(q, t, {k=-1, mode, task}) -> (editDistance, numLocations, endLoc, startLoc, ops)"""
        q, t = _u8(q), _u8(t)
        cap = q.size + t.size + 8
        ops = np.zeros(cap, dtype=np.uint8)
        out = (C.c_int32 * 4)()
        ln = C.c_int32(0)
        self._check(self.lib.dellyhip_edlib_align(self._ctx, _p(q), q.size, _p(t), t.size, mode, task, out,
                                                  ops.ctypes.data_as(C.POINTER(C.c_ubyte)), cap, C.byref(ln)))
        return out[0], out[1], out[2], out[3], ops[:ln.value].tobytes()

    def lcs(self, a, b):
        a, b = _u8(a), _u8(b)
        out = C.c_int32(0)
        self._check(self.lib.dellyhip_lcs(self._ctx, _p(a), a.size, _p(b), b.size, C.byref(out)))
        return out.value

    def gotoh(self, a1, a2):
        r1, m = len(a1), len(a1[0])
        r2, n = len(a2), len(a2[0])
        A1, A2 = _u8(b"".join(a1)), _u8(b"".join(a2))
        cap = m + n + 8
        out = np.zeros((r1 + r2) * cap, dtype=np.uint8)
        ln, score = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.dellyhip_gotoh(self._ctx, _p(A1), r1, m, _p(A2), r2, n, _p(out), cap, C.byref(ln),
                                            C.byref(score)))
        L = ln.value
        return score.value, [out[i * cap:i * cap + L].tobytes() for i in range(r1 + r2)]

    def msa(self, reads):
        """-> (rows, consensus)"""
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        blob = _u8(b"".join(reads))
        cap = int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln, rows = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.dellyhip_msa(self._ctx, len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(cs), cap,
                                          C.byref(ln), C.byref(rows)))
        return rows.value, cs[:ln.value].tobytes()


    def _msa_like(self, fn, reads):
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        blob = _u8(b"".join(reads))
        cap = int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln, rows = C.c_int32(0), C.c_int32(0)
        self._check(fn(self._ctx, len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(cs), cap, C.byref(ln), C.byref(rows)))
        return rows.value, cs[:ln.value].tobytes()

    def msa_wfa(self, reads, prefix=b"", suffix=b""):
        """msaWfa(c, sps, cs, prefix, suffix) -> (rows, consensus)"""
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        blob = _u8(b"".join(reads))
        cap = 2 * int(off[-1]) + 8
        cs = np.zeros(cap, dtype=np.uint8)
        ln, rows = C.c_int32(0), C.c_int32(0)
        pre, suf = _u8(prefix), _u8(suffix)
        self._check(self.lib.dellyhip_msa_wfa(self._ctx, len(reads), _p(blob), _p(off, C.POINTER(C.c_uint64)), _p(pre), pre.size,
                                              _p(suf), suf.size, _p(cs), cap, C.byref(ln), C.byref(rows)))
        return rows.value, cs[:ln.value].tobytes()

    def msa_edlib(self, reads):
        """msaEdlib(c, sps, cs) -> (rows, consensus)"""
        return self._msa_like(self.lib.dellyhip_msa_edlib, reads)


class Stream:
    """dellyhip_stream: the pipelined host-buffer path (the loop of src/shortpe.h:175-201 over many batches).
    submit() returns without waiting; collect() hands out the oldest batch's results."""

    def __init__(self, ctx, depth=3, with_msa=0, want_alignment=False):
        self.ctx = ctx
        self._s = C.c_void_p()
        ctx._check(ctx.lib.dellyhip_stream_create(ctx._ctx, int(depth), int(with_msa), int(bool(want_alignment)), C.byref(self._s)))
        self._keep = []

    def submit(self, batch, tag=0):
        junc = np.ascontiguousarray(batch.junctions)
        blob = _u8(batch.seq_blob)
        off = np.ascontiguousarray(batch.seq_off, dtype=np.uint64)
        self.ctx._check(self.ctx.lib.dellyhip_stream_submit(self._s, int(junc.shape[0]), _p(junc, C.c_void_p), _p(blob),
                                                            _p(off, C.POINTER(C.c_uint64)), C.c_uint64(off.size - 1), C.c_uint64(tag)))

    def zero_copy(self, on=True):
        """dellyhip_stream_zero_copy: pinned sequence bytes are read in place (the caller leaves them alone until the batch is collected)"""
        self.ctx._check(self.ctx.lib.dellyhip_stream_zero_copy(self._s, 1 if on else 0))

    def submit_raw(self, n, junc_ptr, blob_ptr, off_ptr, n_seq, tag=0):
        """pre-marshalled arguments (benchmark loops: no numpy work between the calls)"""
        self.ctx._check(self.ctx.lib.dellyhip_stream_submit(self._s, n, junc_ptr, blob_ptr, off_ptr, n_seq, C.c_uint64(tag)))

    def pending(self):
        return int(self.ctx.lib.dellyhip_stream_pending(self._s))

    def collect(self, copy=True):
        """-> (results, blob, tag); copy=False: views of the stream's pinned block, valid until the next collect()"""
        r, bl = C.c_void_p(), C.c_void_p()
        ln, tag = C.c_uint64(0), C.c_uint64(0)
        n = C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_stream_collect(self._s, C.byref(r), C.byref(bl), C.byref(ln), C.byref(n), C.byref(tag)))
        if n.value == 0:
            if copy:
                self.release()
            return np.zeros(0, dtype=abi.result_dtype()), np.zeros(0, dtype=np.uint8), tag.value
        res = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_uint8)), shape=(n.value * abi.result_dtype().itemsize,)).view(abi.result_dtype())
        blob = np.ctypeslib.as_array(C.cast(bl, C.POINTER(C.c_uint8)), shape=(max(int(ln.value), 1),))[:int(ln.value)]
        if copy:
            res, blob = res.copy(), blob.copy()
            self.release()
        return res, blob, tag.value

    def release(self):
        self.ctx.lib.dellyhip_stream_release(self._s)

    def stats(self, reset=False):
        """host seconds spent staging / launching / enqueueing downloads / waiting, slow-path and top-up batch counts"""
        out = (C.c_double * 6)()
        self.ctx.lib.dellyhip_stream_stats(self._s, out, int(bool(reset)))
        return dict(stage_s=out[0], launch_s=out[1], download_enqueue_s=out[2], wait_s=out[3], slow_batches=int(out[4]), topup_batches=int(out[5]))

    def collect_raw(self):
        """-> (n, blob bytes): no numpy views (benchmark loops)"""
        ln = C.c_uint64(0)
        n = C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_stream_collect(self._s, None, None, C.byref(ln), C.byref(n), None))
        return n.value, ln.value

    def close(self):
        if self._s:
            self.ctx.lib.dellyhip_stream_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentBatch:
    """Junction batch kept in HBM (bench / pipelined callers)."""

    def __init__(self, ctx, batch):
        self.ctx = ctx
        self.n = batch.n
        self._b = C.c_void_p()
        junc = np.ascontiguousarray(batch.junctions)
        blob = _u8(batch.seq_blob)
        self._blob_bytes = int(blob.size) if batch.with_msa != 1 else 0
        off = np.ascontiguousarray(batch.seq_off, dtype=np.uint64)
        rc = ctx.lib.dellyhip_batch_upload(ctx._ctx, self.n, _p(junc, C.c_void_p), _p(blob),
                                           _p(off, C.POINTER(C.c_uint64)), C.c_uint64(off.size - 1),
                                           int(batch.with_msa), C.byref(self._b))
        ctx._check(rc)

    def run(self, stream=None):
        self.ctx._check(self.ctx.lib.dellyhip_batch_run(self.ctx._ctx, self._b, C.c_void_p(stream or 0)))

    def sync(self):
        self.ctx._check(self.ctx.lib.dellyhip_batch_sync(self.ctx._ctx, self._b))

    def kernel_ms(self):
        a, b, l = C.c_double(0), C.c_double(0), C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_kernel_ms(self.ctx._ctx, self._b, C.byref(a), C.byref(b), C.byref(l)))
        return a.value, b.value, l.value

    def dp_kernel_ms(self):
        """Average ms of the dominant alignment kernel alone (split_sparse_kernel, or the packed dense DP kernels when the
        sparse path does not cover the batch) over the launches of the last kernel_ms() window."""
        a = C.c_double(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_dp_kernel_ms(self.ctx._ctx, self._b, C.byref(a)))
        return a.value

    def lr_team_stats(self):
        """long-read batches: (teams launched, junctions they swept in the last run, claims made on the list, error flag)"""
        v = (C.c_int32 * 4)()
        self.ctx._check(self.ctx.lib.dellyhip_batch_lr_team_stats(self.ctx._ctx, self._b, v))
        return tuple(int(x) for x in v)

    def msa_stats(self):
        """msa() batches: (junctions deferred to the direct-float kernel, junctions sent to the second instance, wavefronts per
        junction, blocks) of the last run on this context"""
        v = (C.c_int32 * 4)()
        self.ctx._check(self.ctx.lib.dellyhip_batch_msa_stats(self.ctx._ctx, self._b, v))
        return tuple(int(x) for x in v)

    def sparse_left(self):
        """junctions of the last run that split_sparse_kernel left to the dense kernels"""
        v = C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_sparse_left(self.ctx._ctx, self._b, C.byref(v)))
        return v.value

    def device_results(self):
        """(device pointer, bytes) of the n result records in HBM."""
        ptr, nbytes = C.c_void_p(), C.c_uint64(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_device_results(self.ctx._ctx, self._b, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def fetch_into(self, records, blob):
        """dellyhip_batch_fetch into caller memory: records = uint8 array of >= n * sizeof(record) bytes, blob = uint8 array
        (both e.g. views of a pinned shared-memory segment) -> bytes of blob used"""
        assert records.dtype == np.uint8 and blob.dtype == np.uint8 and records.nbytes >= self.n * abi.result_dtype().itemsize
        used = C.c_uint64(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_fetch(self.ctx._ctx, self._b, C.c_void_p(records.ctypes.data), C.c_void_p(blob.ctypes.data),
                                                          C.c_uint64(blob.nbytes), C.byref(used)))
        return int(used.value)

    def fetch_begin(self, records, blob):
        """dellyhip_batch_fetch_begin: queue the return of this batch's results into PINNED caller memory (uint8 arrays as for
        fetch_into, e.g. the views of a registered shared-memory segment) behind its kernels; does not wait"""
        assert records.dtype == np.uint8 and blob.dtype == np.uint8 and records.nbytes >= self.n * abi.result_dtype().itemsize
        self.ctx._check(self.ctx.lib.dellyhip_batch_fetch_begin(self.ctx._ctx, self._b, C.c_void_p(records.ctypes.data), C.c_void_p(blob.ctypes.data),
                                                                C.c_uint64(blob.nbytes)))

    def fetch_end(self):
        """dellyhip_batch_fetch_end: wait for the fetch begun last -> bytes of blob used"""
        used = C.c_uint64(0)
        try:
            self.ctx._check(self.ctx.lib.dellyhip_batch_fetch_end(self.ctx._ctx, self._b, C.byref(used)))
        except DellyHipError as e:
            e.blob_bytes_needed = int(used.value)
            raise
        return int(used.value)

    def fetch(self):
        res = np.zeros(self.n, dtype=abi.result_dtype())
        cap = self.n * 3100 + 8 * self._blob_bytes + 64
        out = np.zeros(cap, dtype=np.uint8)
        used = C.c_uint64(0)
        self.ctx._check(self.ctx.lib.dellyhip_batch_fetch(self.ctx._ctx, self._b, _p(res, C.c_void_p), _p(out),
                                                          C.c_uint64(cap), C.byref(used)))
        return res, out[:used.value]

    def gather(self, comm, root=0, results_cap=None, blob_cap=None):
        """dellyhip_gather_results: records + consensus / allele bytes of every rank's batch on `root`
        -> (results, blob, counts) there, (None, None, None) elsewhere.  Collective."""
        is_root = comm.rank == root
        ncap = int(results_cap if results_cap is not None else self.n * comm.world + 64)
        bcap = int(blob_cap if blob_cap is not None else (self.n * 3100 + 8 * self._blob_bytes) * comm.world + 64)
        res = np.zeros(ncap if is_root else 0, dtype=abi.result_dtype())
        out = np.zeros(bcap if is_root else 0, dtype=np.uint8)
        counts = np.zeros(comm.world, dtype=np.int32)
        n_res, used = C.c_uint64(0), C.c_uint64(0)
        self.ctx._check(self.ctx.lib.dellyhip_gather_results(
            self.ctx._ctx, comm._c, self._b, int(root), _p(res, C.c_void_p) if is_root else None, C.c_uint64(ncap if is_root else 0),
            C.byref(n_res), _p(out) if is_root else None, C.c_uint64(bcap if is_root else 0), C.byref(used),
            _p(counts, C.POINTER(C.c_int32)) if is_root else None))
        if not is_root:
            return None, None, None
        return res[:n_res.value], out[:used.value], counts

    def gather_into(self, comm, root, pinned):
        """dellyhip_gather_results into preallocated (pinned) host buffers: pinned = (records bytes, blob bytes) as torch uint8
        tensors on the root, None elsewhere -> (records gathered, blob bytes gathered).  Collective."""
        is_root = comm.rank == root
        n_res, used = C.c_uint64(0), C.c_uint64(0)
        if is_root:
            rec, blob = pinned
            ncap = rec.numel() // abi.result_dtype().itemsize
            self.ctx._check(self.ctx.lib.dellyhip_gather_results(
                self.ctx._ctx, comm._c, self._b, int(root), C.c_void_p(rec.data_ptr()), C.c_uint64(ncap), C.byref(n_res),
                C.c_void_p(blob.data_ptr()), C.c_uint64(blob.numel()), C.byref(used), None))
        else:
            self.ctx._check(self.ctx.lib.dellyhip_gather_results(self.ctx._ctx, comm._c, self._b, int(root), None, C.c_uint64(0), C.byref(n_res),
                                                                 None, C.c_uint64(0), C.byref(used), None))
        return n_res.value, used.value

    def gather_device(self, comm, root=0):
        """dellyhip_gather_results_device: the same exchange, results left in the root's HBM -> (records, blob bytes) gathered"""
        dr, db = C.c_void_p(), C.c_void_p()
        n_res, nb = C.c_uint64(0), C.c_uint64(0)
        self.ctx._check(self.ctx.lib.dellyhip_gather_results_device(self.ctx._ctx, comm._c, self._b, int(root), C.byref(dr), C.byref(n_res),
                                                                   C.byref(db), C.byref(nb)))
        return n_res.value, nb.value

    def free(self):
        if self._b:
            self.ctx.lib.dellyhip_batch_free(self.ctx._ctx, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentJobs:
    """AlignJob batch kept in HBM (one process_batch of src/coverage.h:412-450)."""

    def __init__(self, ctx, jobs, blob):
        self.ctx = ctx
        jobs = np.ascontiguousarray(jobs, dtype=abi.align_job_dtype())
        blob = _u8(blob)
        self.n = int(jobs.shape[0])
        self._b = C.c_void_p()
        ctx._check(ctx.lib.dellyhip_jobs_upload(ctx._ctx, C.c_uint64(self.n), _p(jobs, C.c_void_p), _p(blob),
                                                C.c_uint64(blob.size), C.byref(self._b)))

    def run(self, stream=None):
        self.ctx._check(self.ctx.lib.dellyhip_jobs_run(self.ctx._ctx, self._b, C.c_void_p(stream or 0)))

    def sync(self):
        self.ctx._check(self.ctx.lib.dellyhip_jobs_sync(self.ctx._ctx, self._b))

    def kernel_ms(self):
        a, l = C.c_double(0), C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_jobs_kernel_ms(self.ctx._ctx, self._b, C.byref(a), C.byref(l)))
        return a.value, l.value

    def fetch(self):
        res = np.zeros(self.n, dtype=abi.align_result_dtype())
        self.ctx._check(self.ctx.lib.dellyhip_jobs_fetch(self.ctx._ctx, self._b, _p(res, C.c_void_p)))
        return res

    def free(self):
        if self._b:
            self.ctx.lib.dellyhip_jobs_free(self.ctx._ctx, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentNwJobs:
    """_editDistanceNW pairs kept in HBM."""

    def __init__(self, ctx, jobs, blob):
        self.ctx = ctx
        jobs = np.ascontiguousarray(jobs, dtype=abi.nw_job_dtype())
        blob = _u8(blob)
        self.n = int(jobs.shape[0])
        self._b = C.c_void_p()
        ctx._check(ctx.lib.dellyhip_nwjobs_upload(ctx._ctx, C.c_uint64(self.n), _p(jobs, C.c_void_p), _p(blob),
                                                  C.c_uint64(blob.size), C.byref(self._b)))

    def run(self, stream=None):
        self.ctx._check(self.ctx.lib.dellyhip_nwjobs_run(self.ctx._ctx, self._b, C.c_void_p(stream or 0)))

    def kernel_ms(self):
        a, l = C.c_double(0), C.c_int32(0)
        self.ctx._check(self.ctx.lib.dellyhip_nwjobs_kernel_ms(self.ctx._ctx, self._b, C.byref(a), C.byref(l)))
        return a.value, l.value

    def fetch(self):
        out = np.zeros(self.n, dtype=np.int32)
        self.ctx._check(self.ctx.lib.dellyhip_nwjobs_fetch(self.ctx._ctx, self._b, _p(out, C.c_void_p)))
        return out

    def free(self):
        if self._b:
            self.ctx.lib.dellyhip_nwjobs_free(self.ctx._ctx, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
