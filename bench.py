#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: candidate split-read alignments/sec
(DEL, 150 bp consensus, 1 kb reference window) on N MI355X.

A "step" is one pass of the hot path (alignConsensus: window construction,
longNeedle, split detection, coordinates) over one resident batch of synthetic
junctions (BASELINE config 2: 10 000 junctions per GPU); the steps
rotate through RESIDENT_BATCHES different batches and, at N = 1, alternate between
two contexts on two HIP streams, so two launches are in flight and the tail of
one runs under the head of the next (`roofline.one_launch_at_a_time` has the
figures of a launch that has the chip to itself).  Inputs are in HBM before the
timed region and result records stay in HBM: that is `value` (the bench
contract).  The quantity SURVEY.md 8d defines -- host buffers in, host buffers out,
marshalling + H2D + kernels + D2H -- is measured in the same run through the
pipelined path (dellyhip_stream) over >= 1 s of batches and reported beside it as
`host_inclusive`.  For N > 1 junctions shard across ranks and the results of
every step reach host memory rank 0 can read, inside the timed region, in two
ways that are timed one after the other: `value` with the pipelined per-rank
return (dellyhip_batch_fetch_begin / _end: step k queues the return of step k - 1
behind its kernels, a kernel writes records and bytes over the rank's own PCIe
link into a pinned POSIX shared-memory segment rank 0 has mapped; no collective,
two launches in flight as at N = 1), and `config.gather_*` with the blocking gather
to rank 0 (dellyhip_gather_results over RCCL, then one download there).
`--gather shm` / `--gather rccl` time one of them only.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task description) with the
`roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_U = 1214        # SURVEY.md 8d / BASELINE.md 4: m + n + 64 B record at C2
CELLS_PER_U = 2 * 151 * 1001  # fwd + rev DP cells
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


# The side measurements of side_measurements(): (name, junctions, CPU sample, batch recipe).  The recipe is synth.make_batch's
# keyword arguments plus bench-only keys (leading underscore, see side_batch()): _tiles = that many copies of the batch side by
# side (chip-filling variants), _big = (flank, DEL length) of synth.make_big_deletions, _steps = timed steps, _cpu_threads = cap
# on the reference's threads, _no_stream = no host-inclusive leg.
# tests/test_gpu_bench_shapes.py bit-compares the HIP path with oracle/_ref on exactly these batches.
SIDE_PLAN = (("u_full_n20", 2000, 2000, dict(mode="c2", n_reads=20)),
             ("u_full_n20_10k_junctions", 10000, 0, dict(mode="c2", n_reads=20)),   # the chip filled: one wavefront per junction needs > 4 096 of them
             ("u_full_n5", 2000, 2000, dict(mode="c2", n_reads=5)),
             # BASELINE configs[2] as written: "all SV types, full sr pipeline" -- svt 0 .. 8 (translocations on a second
             # chromosome, insertions through splitAlign), 2 .. 20 reads per junction: msa + alignConsensus
             ("sr_stage_mixed_all_svt", 10000, 2000, dict(mode="allsvt", n_reads=(2, 20))),
             ("ins_svt4", 5000, 5000, dict(mode="ins")),
             # The order matters, for reasons the builder could not isolate (tools/ctx_reuse.py, ctx_reuse2.py reproduce nothing): rows of
             # short kernels measured AFTER this row and its five-slot stream lose ~0.5 ms per step (u_full_n20: 2.1 instead of 1.6 ms,
             # same kernel times), and this row measured after ALL the long-read rows reads 41 instead of 53 M/s.  So: the short
             # msa() rows first, then this one, then the long-read rows (tens of milliseconds per step: not sensitive).
             ("u_c2_40k_junctions", 40000, 0, dict(mode="c2")),
             ("lr_c4_align_consensus", 2048, 128, dict(mode="lr", sub_rate=0.01)),
             ("lr_c4_msaedlib_n15", 768, 64, dict(mode="lr", n_reads=15, sub_rate=0.06)),
             # INS 800 bp, 15 reads of ~3.8 kb at 6 % error: msaWfa + alignConsensus (splitAlign)
             ("lr_ins_msawfa_n15", 512, 64, dict(mode="lrins", n_reads=15, sub_rate=0.06)),
             # BASELINE configs[3] as written (SURVEY.md F6 "benchmark both"): 10 kb consensus x 20.7 kb window; the reference
             # holds four int32 matrices of 830 MB per call (src/needle.h:52-103): 8 threads at most
             ("lr_stress_10kb_x_20kb", 64, 8, dict(mode="lr", _big=(5000, 700), _cpu_threads=8, _steps=2, _no_stream=True)),
             # chip-filling sizes of the long-read rows (one wavefront per junction: at the sizes above the kernels are bound by a
             # junction's latency, these show the throughput with every wavefront slot busy): four tiles of the row's batch, one step
             ("lr_c4_align_consensus_8k", 8192, 0, dict(mode="lr", sub_rate=0.01, _tiles=4, _steps=1, _no_stream=True)),
             ("lr_c4_msaedlib_n15_3k", 3072, 0, dict(mode="lr", n_reads=15, sub_rate=0.06, _tiles=4, _steps=1, _no_stream=True)),
             ("lr_ins_msawfa_n15_2k", 2048, 0, dict(mode="lrins", n_reads=15, sub_rate=0.06, _tiles=4, _steps=1, _no_stream=True)))
SIDE_PLAN_BIG = ()   # (round 4 kept the chip-filling rows out of the default run; they are tiles now and part of it)


def side_batch(synth, n, kw):
    """the batch of a SIDE_PLAN row"""
    kw = dict(kw)
    tiles = int(kw.pop("_tiles", 1))
    big = kw.pop("_big", None)
    for k in [k for k in kw if k.startswith("_")]:
        kw.pop(k)
    if big is not None:
        return synth.make_big_deletions([tuple(big)] * n, seed=23, err=0.01, revcomp_every=3)
    b = synth.make_batch(n // tiles, **kw)
    return synth.tile_batch(b, tiles)


RESIDENT_BATCHES = 4          # distinct resident batches the timed steps rotate through
MULTI_RESIDENT_BATCHES = 4   # N > 1: a batch is run again four steps later, long after its return (queued one step after its run) has left
HOST_INCLUSIVE_SECONDS = 1.0  # wall time of the pipelined host-buffer measurement
STREAM_DEPTH = 6

# deficit sweep: the C2 shape under different consensus / genome content.  The sparse longNeedle's cost
# grows with a junction's deficit (errors between consensus and reference); the reference's does not (src/needle.h:64-115
# fills every cell).  (name, synth.make_batch kwargs); tests/test_gpu_bench_shapes.py bit-compares every one with oracle/_ref.
SWEEP_PLAN = (("substitutions_0", dict(sub_rate=0.0)),
              ("substitutions_0.5pct_baseline", dict()),
              ("substitutions_2pct", dict(sub_rate=0.02)),
              ("substitutions_5pct", dict(sub_rate=0.05)),
              ("indel_1to3bp_per_consensus", dict(read_indel=1.0)),
              ("nontemplated_insertion_12bp", dict(junction_ins=12)),
              ("real_chr18_windows", dict(genome="real")),
              ("low_complexity_mix", dict(genome="lowcx")))
SWEEP_N = 10000


def sweep_batch(synth, kw, n=SWEEP_N):
    kw = dict(kw)
    if kw.get("genome"):
        kw["real"] = synth.load_real_chromosome()
    return synth.make_batch(n, mode="c2", seed=77, **kw)


class _DevPtr:
    """Wraps a raw device pointer for torch.as_tensor (RCCL needs a tensor)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def cpu_baseline(batch, budget_s=12.0):
    """The reference's CPU path (oracle/_ref: its own headers) or the C port, timed on this box's host cores with
    the reference's threading model (src/shortpe.h:175-201: std::threads on one atomic counter) over the SAME
    10 000 C2 junctions.  Only the loop body is timed -- ONE alignConsensus() per junction, no diagnostic replay,
    no marshalling -- and the clock runs inside the C++ driver around thread start .. join
    (oracle/ref_driver.cpp: dref_time_refine_batch).  Every thread gets >= 32 junctions per pass.

    The thread count is SWEPT (nproc/4, nproc/2, nproc, each ~budget_s/6 of passes) and `value` / `cores` are the BEST
    of them: on the 256-thread boxes of this pool all threads are NOT the fastest -- every longNeedle
    call allocates and frees four (m+1)(n+1) int32 matrices (0.6 MB each at C2, src/needle.h:52-103), which glibc
    serves with mmap / munmap above its 128 KB threshold, and those calls serialise on the process's mmap lock."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    orc = pyoracle.Oracle(kind)
    cores = os.cpu_count() or 1
    # single thread: ~1 s on a 512-junction prefix
    cal = _subbatch(batch, 512)
    s1, n1, _ = orc.time_refine(cal, n_threads=1, reps=1)
    rate1 = n1 / s1
    cap = max(1, min(cores, batch.n // 32))     # >= 32 junctions per thread and pass
    counts = sorted({max(1, min(cap, c)) for c in (cores // 4, cores // 2, cores)})
    sweep = []
    for threads in counts:   # a short leg per thread count (the first pass also warms the thread stacks / page cache)
        s0, _, _ = orc.time_refine(batch, n_threads=threads, reps=1)
        reps = int(max(1, min(100, budget_s / 6.0 / max(s0, 1e-3))))
        sN, nN, okN = orc.time_refine(batch, n_threads=threads, reps=reps)
        sweep.append({"cores": threads, "value": nN / sN, "seconds": sN, "passes": reps})
    best = max(sweep, key=lambda r: r["value"])
    threads = best["cores"]
    reps = int(max(1, min(400, budget_s / 2.0 * best["value"] / batch.n)))
    sN, nN, okN = orc.time_refine(batch, n_threads=threads, reps=reps)
    if nN / sN < best["value"]:      # (the longer leg is the figure unless the sweep's own leg was faster)
        sN, nN = best["seconds"], int(round(best["value"] * best["seconds"]))
        reps = best["passes"]
    return {"value": nN / sN, "unit": "alignments/s", "cores": threads, "kind": kind,
            "value_one_thread": rate1, "host_threads": cores,
            "thread_sweep": [{"cores": r["cores"], "value": r["value"]} for r in sweep],
            "sample": "best of the thread counts %s (alignments/s: %s); %d passes over the same %d C2 junctions (%d alignConsensus calls, %d returned "
                      "true) on %d std::thread workers pulling from one atomic counter (src/shortpe.h:175-201 model), %.1f s measured inside the "
                      "C++ driver around thread start..join; one thread: %d junctions in %.2f s.  More threads than that are slower: every "
                      "longNeedle call mmaps and unmaps four 0.6 MB int32 matrices (src/needle.h:52-103), serialised by the process's mmap lock"
                      % ([r["cores"] for r in sweep], [int(r["value"]) for r in sweep], reps, batch.n, nN, okN, threads, sN, n1, s1)}


def _profile_file(name):
    """newest committed copy of a profile artefact (profiles/rNN/<name>)"""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            return path
    return None


def _measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -> dict(traffic, valu, stale, source); traffic None if not
    collected.  The file carries the hash of the kernel's sources at profiling time (tools/pmc_traffic_summary.py); `stale` says
    whether the tree this bench runs from still has those sources -- a file without a stamp is stale."""
    out = {"traffic": None, "valu": None, "stale": None, "source": None}
    try:
        path = _profile_file("pmc_traffic.json")
        with open(path) as f:
            j = json.load(f)
        from delly_amd import build as dbuild
        out["traffic"] = j["hbm_bytes_per_launch"]
        out["valu"] = j.get("valu_wave_instructions_per_launch")
        out["stale"] = j.get("kernel_source_sha16") != dbuild.headline_kernel_hash()
        out["source"] = os.path.relpath(path, ROOT)
    except Exception:
        pass
    return out


# VALU issue ceiling (profiles/r03/valu_clock.txt, tools/valu_clock.hip: the shader clock is MEASURED there -- s_memtime
# against the 100 MHz s_memrealtime: 2.35 GHz under load, rocm-smi agrees): a SIMD issues a wave64 integer VALU instruction
# every ~2 cycles (v_add_u32 1.5, v_fma_f32 1.9, v_max_i32 / v_pk_add_i16 2.5) once >= 8 wavefronts are resident, as
# MI355X_MICROARCH.md says; ONE wavefront alone issues one every 5.8 cycles, so a kernel with W resident wavefronts per
# SIMD cannot exceed min(W / 5.8, ~0.5) instructions per cycle.  Round 2's "0.55 G/s measured ceiling" was a 2-4-wave figure.
VALU_CEILING = {"cycles_per_wave64_valu_at_8_waves_per_simd": 2.0, "cycles_per_instruction_one_wave": 5.8, "sclk_mhz_measured": 2350,
                "source": "profiles/r03/valu_clock.txt (tools/valu_clock.hip)"}


def _subbatch(batch, n):
    from delly_amd import synth
    n = min(n, batch.n)
    first = int(batch.junctions["seq_first"][0])
    last = int(batch.junctions["seq_first"][n - 1] + batch.junctions["n_seq"][n - 1])
    return synth.Batch(batch.chroms, batch.junctions[:n].copy(), batch.seq_blob, batch.seq_off[:last + 1].copy(),
                       batch.with_msa, batch.truth[:n] if batch.truth is not None else None)


def one_genome(synth, batches):
    """independently generated batches -> one chromosome table (concatenated), coordinates shifted"""
    import numpy as np
    chroms = [np.concatenate([b.chroms[c] for b in batches]) for c in range(len(batches[0].chroms))]
    out, base = [], [0] * len(chroms)
    for b in batches:
        j = b.junctions.copy()
        j["sv_start"] += base[0]
        j["sv_end"] += np.where(j["chr2"] == 0, base[0], base[-1])
        out.append(synth.Batch(chroms, j, b.seq_blob, b.seq_off, b.with_msa, b.truth))
        base = [x + c.size for x, c in zip(base, b.chroms)]
    return chroms, out


def host_inclusive_rate(ctx, batches, with_msa, seconds=HOST_INCLUSIVE_SECONDS, depth=STREAM_DEPTH, min_batches=3, pinned_input=False):
    """host buffers in, host buffers out.  The batches (host arrays) cycle through a dellyhip_stream with `depth`
    slots -- depth - 1 in flight while the consumer still holds the block of the last collect -- for >= `seconds` of wall
    time; the clock covers validation, routing, staging copies, H2D, kernels, device-side compaction, D2H and the waits."""
    import numpy as np
    from delly_amd import abi, refine
    st = refine.Stream(ctx, depth=depth, with_msa=with_msa)
    args = []
    registered = []
    if pinned_input:
        # the caller keeps its sequence bytes in pinned memory and leaves them alone until the batch is collected
        # (dellyhip_stream_zero_copy): the copy engine reads them in place, no staging copy in submit
        st.zero_copy(True)
    for b in batches:
        junc = np.ascontiguousarray(b.junctions)
        blob = np.ascontiguousarray(b.seq_blob, dtype=np.uint8)
        if pinned_input:
            blob = blob.copy()
            ctx.host_register(blob.ctypes.data, blob.nbytes)
            registered.append(blob)
        off = np.ascontiguousarray(b.seq_off, dtype=np.uint64)
        args.append((junc.shape[0], junc.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_char_p),
                     off.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(off.size - 1), (junc, blob, off)))
    state = {"nxt": 0, "k": 0}

    def pump(until_batches=None, until_time=None):
        nj = nb = 0
        inflight = max(1, depth - 1)
        while True:
            if until_batches is not None and state["k"] >= until_batches:
                break
            if until_time is not None and time.perf_counter() >= until_time and state["k"] >= min_batches:
                break
            if depth == 1:
                st.release()
            while state["nxt"] - state["k"] < inflight:
                a = args[state["nxt"] % len(args)]
                st.submit_raw(a[0], a[1], a[2], a[3], a[4], state["nxt"])
                state["nxt"] += 1
            n, ln = st.collect_raw()
            state["k"] += 1
            nj += n
            nb += ln
        return nj, nb

    pump(until_batches=max(depth + 1, len(args)))          # warm-up: the buffers grow to the batch size
    while st.pending():                                    # drain, so that the timed region starts and ends idle
        st.collect_raw()
        state["k"] += 1
    state["nxt"] = state["k"] = 0
    st.stats(reset=True)
    t0 = time.perf_counter()
    nj, nb = pump(until_time=t0 + seconds)
    while st.pending():
        n, ln = st.collect_raw()
        state["k"] += 1
        nj += n
        nb += ln
    dt = time.perf_counter() - t0
    total = state["k"]
    stats = st.stats()
    st.close()
    for blob in registered:
        ctx.host_unregister(blob.ctypes.data)
    up = sum(a[5][0].nbytes + a[5][1].nbytes + a[5][2].nbytes for a in args) / len(args)
    return {"value": nj / dt, "unit": "junctions/s", "batches": total, "junctions_per_batch": nj / max(total, 1), "wall_s": dt,
            "ms_per_batch": dt / max(total, 1) * 1e3, "depth": depth,
            "host_ms_per_batch": {k: (v / max(total, 1) * 1e3 if k.endswith("_s") else v) for k, v in stats.items()},
            "bytes_up_per_batch": int(up), "bytes_down_per_batch": int(nb / max(total, 1) + nj / max(total, 1) * abi.result_dtype().itemsize),
            "note": "dellyhip_stream: host buffers in -> host buffers out (validation, routing, pinned staging, H2D, kernels, compaction, D2H); chromosome resident"}


def host_inclusive_rate_threads(ctx, batches, with_msa, threads, seconds=HOST_INCLUSIVE_SECONDS, depth=4, pinned_input=True):
    """The same loop from `threads` caller threads at once, each with a dellyhip_stream of its own on `ctx` (the reference's callers are
    the workers of a thread pool, src/shortpe.h:80,175-201: c.maxThreads = 4 by default): a submit's host work -- validation, window
    lengths, routing, staging the records -- is ~0.13 ms per 10 000 junctions on one thread, more than the kernels take.  ctypes
    releases the GIL inside the calls."""
    import threading
    out = [None] * threads
    err = []
    start = threading.Barrier(threads)

    def run(k):
        try:
            start.wait()
            out[k] = host_inclusive_rate(ctx, batches[k % len(batches):] + batches[:k % len(batches)], with_msa, seconds=seconds, depth=depth,
                                         pinned_input=pinned_input)
        except Exception as e:   # pragma: no cover
            err.append(repr(e))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    if err or any(o is None for o in out):
        return {"error": "; ".join(err) or "a caller thread returned nothing"}
    nj = sum(o["value"] * o["wall_s"] for o in out)
    return {"value": sum(o["value"] for o in out), "unit": "junctions/s", "threads": threads, "depth_per_thread": depth,
            "per_thread": [o["value"] for o in out], "ms_per_batch_per_thread": [o["ms_per_batch"] for o in out],
            "junctions": nj, "wall_s_incl_warmup": wall,
            "note": "sum of the threads' own rates (each over its own >= %.1f s timed region, all regions concurrent)" % seconds}


def deficit_sweep(ctx, synth, device, with_cpu, steps=5):
    """SWEEP_PLAN: the headline shape under other consensus / genome content.  Per point: resident alignments/s, how many
    junctions the sparse kernel resolved per level bucket (deficit = |consensus| - bestScore of the resolved junctions), how
    many it left to the dense kernels, and the reference's CPU rate on a 1000-junction prefix (which should not move)."""
    import numpy as np
    out = {}
    orc = None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        orc = pyoracle.Oracle("reference" if pyoracle.have_reference() else "port")
    cores = os.cpu_count() or 1
    for name, kw in SWEEP_PLAN:
        b = sweep_batch(synth, kw)
        ctx.set_chromosomes(b.chroms)
        rb = ctx.upload(b)
        rb.run(); rb.sync(); rb.kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            rb.run()
        rb.sync()
        dt = (time.perf_counter() - t0) / steps
        ms_split, _, _ = rb.kernel_ms()
        ms_dp = rb.dp_kernel_ms()
        left = rb.sparse_left()
        res, _ = rb.fetch()
        ran = (res["status"] == 0) & (res["score_best"] != -1)
        deficit = (res["cons_len"] - res["score_best"])[ran]
        edges = [0, 2, 4, 6, 8, 16, 32]
        hist = {}
        lo = -1
        for e in edges:
            hist["<=%d" % e] = int(((deficit > lo) & (deficit <= e)).sum())
            lo = e
        hist[">32"] = int((deficit > 32).sum())
        out[name] = {"junctions": b.n, "alignments_per_s": b.n / dt, "ms_per_step": dt * 1e3, "sparse_kernel_ms": ms_dp, "all_split_kernels_ms": ms_split,
                     "refined_ok": int(res["ok"].sum()), "left_to_dense_kernels": int(left), "left_to_dense_frac": left / max(b.n, 1),
                     "deficit_histogram": hist, "make_batch": {k: v for k, v in kw.items()}}
        rb.free()
        if orc is not None:
            sub = _subbatch(b, 1000)
            threads = max(1, min(cores, sub.n // 8))
            sec, visits, _ = orc.time_refine(sub, n_threads=threads, reps=1)
            reps = int(max(1, min(20, 1.0 / max(sec, 1e-3))))
            if reps > 1:
                sec, visits, _ = orc.time_refine(sub, n_threads=threads, reps=reps)
            out[name]["cpu_" + orc.kind] = {"alignments_per_s": visits / sec, "cores": threads, "sample": "%d x %d junctions, %.2f s" % (reps, sub.n, sec)}
    return out


def side_measurements(ctx, synth, device=0, steps=3, with_cpu=True, only=None):
    """Not the headline: what `delly sr` / `delly lr` pay per junction beyond unit U -- msa of N reads +
    alignConsensus (U_full), the insertion path (splitAlign/edlib) and the long-read shapes
    of BASELINE config C4 -- each over a resident batch, whole-step wall clock.  with_cpu: the same
    workloads through the CPU checker (oracle/_ref = the reference's own code, else the C port) on a small
    bounded sample with all host threads; this is part of bench.py's cpu_baseline leg."""
    from delly_amd import abi, refine
    orc = None
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        orc = pyoracle.Oracle("reference" if pyoracle.have_reference() else "port")
    cores = os.cpu_count() or 1
    out = {}
    want = (lambda name: True) if not only else (lambda name: name in only)
    plan = tuple(x for x in SIDE_PLAN if want(x[0]))
    ctx_sr, ctx_lr = ctx, None
    for name, n, ncpu, kw in plan:
        b = side_batch(synth, n, kw)
        lr = kw["mode"].startswith("lr")
        row_steps = int(kw.get("_steps", steps))
        params = abi.params_lr(realign=True) if lr else abi.params_sr()
        if lr:   # long-read parameters + orientation test (src/tegua.h:237-241, src/assemble.h:849): a context of their own
            if ctx_lr is None:
                ctx_lr = refine.Context(params=params, device=device)
            ctx = ctx_lr
        else:
            ctx = ctx_sr
        if os.environ.get("BENCH_TRIM_BETWEEN_ROWS"):
            ctx.trim_memory()
        ctx.set_chromosomes(b.chroms)
        rb = ctx.upload(b)
        rb.run(); rb.sync(); rb.kernel_ms()
        def region(k):   # k steps behind one another, one wait: (seconds per step, ms the first enqueue took)
            t0 = time.perf_counter()
            rb.run()
            t1 = time.perf_counter()
            for _ in range(k - 1):
                rb.run()
            rb.sync()
            return (time.perf_counter() - t0) / k, (t1 - t0) * 1e3
        # A side row is the MEDIAN of three regions: a region that follows the tear-down of another row's pipelined stream can
        # lose 0.4 - 25 ms -- the first enqueue behind a sync returns late and the GPU finishes late, with the old library as
        # with the new one, and none of it inside dellyhip_batch_run's own clock (BENCH_ROW_DEBUG; CHANGELOG round 6, the same
        # family as VERDICT r05 #7's "0.5 ms per step a five-slot stream leaves behind").  Regions shorter than ~25 ms are
        # stretched to ~50 ms of steps so that what is left of such a delay does not carry weight.
        regs = [region(row_steps)]
        if regs[0][0] * row_steps < 0.025 and not kw.get("_steps"):
            row_steps = int(min(200, max(row_steps, round(0.05 / max(regs[0][0], 1e-5)))))
            rb.kernel_ms()
            regs = [region(row_steps)]
        regs += [region(row_steps) for _ in range(2)]
        dt = sorted(r[0] for r in regs)[1]
        first_enqueue_ms = max(r[1] for r in regs)
        if os.environ.get("BENCH_ROW_DEBUG"):
            print("row %s: %d steps per region; ms per step %s; first enqueue ms %s" % (name, row_steps, ["%.3f" % (r[0] * 1e3) for r in regs], ["%.3f" % r[1] for r in regs]),
                  file=sys.stderr, flush=True)
        ms_split, ms_msa, _ = rb.kernel_ms()
        res, _ = rb.fetch()
        out[name] = {"junctions": n, "junctions_per_s": n / dt, "ms_per_step": dt * 1e3, "msa_stage_ms": ms_msa,
                     "split_stage_ms": ms_split, "refined_ok": int(res["ok"].sum()), "steps": row_steps,
                     "regions_ms_per_step": [r[0] * 1e3 for r in regs], "first_enqueue_ms_max": first_enqueue_ms}
        if b.with_msa == 1:   # short-read msa(): how many junctions left the score-table kernel (CHANGELOG.md 4)
            try:
                ms = rb.msa_stats()
                out[name].update({"msa_deferred_junctions": ms[0], "msa_second_instance_junctions": ms[1], "msa_wavefronts_per_junction": ms[2]})
            except Exception:
                pass
        if lr:   # the dense strips of junctions the sparse passes give up on run on teams of wavefronts (CHANGELOG.md 3.7)
            try:
                ts = rb.lr_team_stats()
                out[name]["lr_teams"], out[name]["lr_team_junctions"] = ts[0], ts[1]
            except Exception:
                pass
        rb.free()
        try:   # the same batch from host buffers through the pipelined path
            if kw.get("_no_stream"):
                raise KeyError("skipped")
            hi = host_inclusive_rate(ctx, [b], b.with_msa, seconds=0.5 if dt < 0.05 else 3 * dt, depth=5 if dt < 0.02 else 3)
            out[name]["host_inclusive"] = {k: hi[k] for k in ("value", "unit", "batches", "wall_s", "ms_per_batch", "depth", "bytes_up_per_batch", "bytes_down_per_batch")}
            out[name]["host_inclusive"]["vs_resident"] = hi["value"] / (n / dt)
        except KeyError:
            pass
        except Exception as e:
            out[name]["host_inclusive"] = {"error": repr(e)}
        if orc is not None and ncpu > 0:
            sub = b if ncpu >= n else _subbatch(b, ncpu)   # (a prefix of the SAME batch)
            threads = max(1, min(cores, sub.n, int(kw.get("_cpu_threads", cores))))
            sec, visits, _ = orc.time_refine(sub, n_threads=threads, reps=1, params=params)
            reps = int(max(1, min(40, 1.5 / max(sec, 1e-3))))   # ~1.5 s of CPU work per workload
            if reps > 1:
                sec, visits, _ = orc.time_refine(sub, n_threads=threads, reps=reps, params=params)
            out[name]["cpu_" + orc.kind] = {"junctions_per_s": visits / sec, "cores": threads,
                                            "sample": "%d x %d junctions, %.2f s inside the C++ driver" % (reps, sub.n, sec)}
    # the split-read genotyping classifier (src/coverage.h:412-434), one process_batch of
    # 131072 x 8 AlignJobs (:271) resident in HBM: 26..37-byte probes against 150-byte reads
    try:
        if not want("sr_genotype_classifier"):
            raise KeyError("skipped")
        import numpy as np
        base_jobs, base_blob = synth.make_align_jobs(160, 40, seed=9)
        tiles = (131072 * 8 + base_jobs.shape[0] - 1) // base_jobs.shape[0]
        jobs = np.tile(base_jobs, tiles)
        shift = np.repeat(np.arange(tiles, dtype=np.uint64) * np.uint64(base_blob.size), base_jobs.shape[0])
        for f in ("cons_off", "ref_off", "seq_off"):
            jobs[f] += shift
        blob = np.tile(base_blob, tiles)
        cx = refine.Context(device=device)
        rj = refine.ResidentJobs(cx, jobs, blob)
        rj.run(); rj.sync(); rj.kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            rj.run()
        rj.sync()
        dt = (time.perf_counter() - t0) / steps
        kms, _ = rj.kernel_ms()
        res = rj.fetch()
        nj = int(jobs.shape[0])
        cells = float((jobs["cons_len"].astype(np.int64) + jobs["ref_len"]).astype(np.float64) @ jobs["seq_len"].astype(np.float64))
        alg_bytes = float(jobs["cons_len"].sum() + jobs["ref_len"].sum() + jobs["seq_len"].sum()) + nj * (48 + 20)
        out["sr_genotype_classifier"] = {
            "jobs": nj, "jobs_per_s": nj / dt, "ms_per_step": dt * 1e3, "classify_kernel_ms": kms,
            "gcups": cells / (kms * 1e-3) / 1e9, "hbm_frac": alg_bytes / (kms * 1e-3) / 8e12,
            "types": {t: int((res["type"] == ord(t)).sum()) for t in "RAN"},
            "note": "two edlib HW distances per job (probe x read); one job per lane, 64-bit Myers"}
        rj.free()
        cx.close()
        if orc is not None:
            sub = slice(0, 40 * base_jobs.shape[0])
            ref = orc.classify_reads(jobs[sub], blob, n_threads=cores, with_dist=False)
            dtc = orc.worker_seconds   # the thread-pool region of process_batch alone
            same = all((ref[f] == res[sub][f]).all() for f in ("type", "qual", "sv_id", "file_index"))
            out["sr_genotype_classifier"]["cpu_" + orc.kind] = {"jobs_per_s": ref.shape[0] / dtc, "cores": cores,
                                                                "sample": "%d jobs, %.2f s" % (ref.shape[0], dtc),
                                                                "identical_to_gpu": bool(same)}
    except KeyError:
        pass
    except Exception as e:  # side figure only
        out["sr_genotype_classifier"] = {"error": repr(e)}
    # long-read genotyping, _editDistanceNW (src/genotype.h:21-30,276,284): read slice vs REF and
    # ALT slices of 1000..2000 bytes, 6 % ONT-like error
    try:
        if not want("lr_genotype_edit_distance_nw"):
            raise KeyError("skipped")
        import numpy as np
        base_jobs, base_blob = synth.make_nw_jobs(512, seed=19)
        tiles = 16
        jobs = np.tile(base_jobs, tiles)
        shift = np.repeat(np.arange(tiles, dtype=np.uint64) * np.uint64(base_blob.size), base_jobs.shape[0])
        for f in ("query_off", "target_off"):
            jobs[f] += shift
        blob = np.tile(base_blob, tiles)
        cx = refine.Context(device=device)
        rj = refine.ResidentNwJobs(cx, jobs, blob)
        rj.run(); rj.fetch(); rj.kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            rj.run()
        dist = rj.fetch()
        dt = (time.perf_counter() - t0) / steps
        kms, _ = rj.kernel_ms()
        nj = int(jobs.shape[0])
        cells = float(jobs["query_len"].astype(np.float64) @ jobs["target_len"].astype(np.float64))
        out["lr_genotype_edit_distance_nw"] = {"pairs": nj, "pairs_per_s": nj / dt, "ms_per_step": dt * 1e3,
                                               "nw_jobs_kernel_ms": kms, "gcups": cells / (kms * 1e-3) / 1e9,
                                               "mean_len": float(jobs["query_len"].mean())}
        rj.free()
        cx.close()
        if orc is not None:
            sub = slice(0, nj)
            dtc, ncpu = 0.0, 0
            while dtc < 0.5 and ncpu < 64:   # >= 0.5 s of worker time: one pass is only tens of milliseconds
                ref = orc.edit_distance_nw_batch(jobs[sub], blob, n_threads=cores)
                dtc += orc.worker_seconds
                ncpu += 1
            dtc /= ncpu
            orc.edit_distance_nw_batch(jobs[:2048], blob, n_threads=1)
            dt1 = orc.worker_seconds * 256 / 2048
            out["lr_genotype_edit_distance_nw"]["cpu_" + orc.kind] = {
                "pairs_per_s": ref.shape[0] / dtc, "cores": cores, "sample": "%d x %d pairs, %.3f s each" % (ncpu, ref.shape[0], dtc),
                "pairs_per_s_one_thread": 256 / dt1, "identical_to_gpu": bool((ref == dist[sub]).all()),
                "note": "the reference calls _editDistanceNW serially per read (src/genotype.h:262-284)"}
    except KeyError:
        pass
    except Exception as e:  # side figure only
        out["lr_genotype_edit_distance_nw"] = {"error": repr(e)}
    return out


# (side row, key of its rate in `config`)
FLAT_ROWS = (("u_c2_40k_junctions", "u_c2_40k_alignments_per_s"), ("u_full_n20", "u_full_n20_2k_junctions_per_s"),
             ("u_full_n20_10k_junctions", "u_full_n20_10k_junctions_per_s"), ("u_full_n5", "u_full_n5_2k_junctions_per_s"),
             ("sr_stage_mixed_all_svt", "sr_stage_mixed_all_svt_junctions_per_s"), ("ins_svt4", "ins_svt4_junctions_per_s"),
             ("lr_c4_align_consensus", "lr_c4_align_consensus_junctions_per_s"), ("lr_c4_msaedlib_n15", "lr_c4_msaedlib_n15_junctions_per_s"),
             ("lr_ins_msawfa_n15", "lr_ins_msawfa_n15_junctions_per_s"), ("lr_stress_10kb_x_20kb", "lr_stress_10kb_x_20kb_junctions_per_s"),
             ("lr_c4_align_consensus_8k", "lr_c4_align_consensus_8k_junctions_per_s"), ("lr_c4_msaedlib_n15_3k", "lr_c4_msaedlib_n15_3k_junctions_per_s"),
             ("lr_ins_msawfa_n15_2k", "lr_ins_msawfa_n15_2k_junctions_per_s"))
# The driver's record of a run keeps the first DRIVER_CONFIG_KEYS scalar entries of `config`: these come first, in this order
# (tests/test_bench_record.py parses a line the way the driver does and asserts they survive).
DRIVER_CONFIG_KEYS = 24
CONFIG_FIRST = ("workload", "value_is", "one_launch_at_a_time_alignments_per_s", "one_launch_at_a_time_kernel_ms", "host_inclusive_alignments_per_s",
                "host_inclusive_full_payload_alignments_per_s", "u_full_n20_10k_junctions_per_s", "u_full_n20_2k_junctions_per_s", "sr_stage_mixed_all_svt_junctions_per_s",
                "ins_svt4_junctions_per_s", "lr_c4_align_consensus_junctions_per_s", "lr_c4_msaedlib_n15_junctions_per_s",
                "lr_ins_msawfa_n15_junctions_per_s", "lr_stress_10kb_x_20kb_junctions_per_s", "lr_c4_align_consensus_8k_junctions_per_s",
                "lr_c4_msaedlib_n15_3k_junctions_per_s", "lr_ins_msawfa_n15_2k_junctions_per_s",
                "substitutions_2pct_alignments_per_s", "substitutions_5pct_alignments_per_s",
                "u_c2_40k_alignments_per_s", "u_full_n5_2k_junctions_per_s",
                "value_min", "value_max", "refined_ok_min")
# N > 1 (the driver's SCALE runs): what the return paths cost comes first
CONFIG_FIRST_MULTI = ("workload", "value_is", "value_return_path", "gather_alignments_per_s", "gather_step_ms", "gather_ms_per_step", "gather_transport", "rccl_ranks",
                      "shm_return_alignments_per_s", "shm_return_ms_per_step", "shm_return_gather_ms_per_step", "host_inclusive_alignments_per_s", "ms_per_step_min_rank", "ms_per_step_max_rank",
                      "ranks_launched", "ranks_that_ran_kernels", "oversubscribed_one_device", "gathered_records_on_rank0",
                      "gathered_blob_bytes_on_rank0", "shm_return_records_seen_by_rank0", "junctions_per_gpu", "refined_ok_min",
                      "kernels_ms_per_step_rank0", "launches_in_flight")


def order_config(cfg):
    first = CONFIG_FIRST_MULTI if "value_return_path" in cfg else CONFIG_FIRST
    out = {k: cfg[k] for k in first if k in cfg}
    out.update({k: v for k, v in cfg.items() if k not in out})
    return out


def driver_view_of_config(cfg, keep=DRIVER_CONFIG_KEYS):
    """what BENCH_rNN.json keeps of `config`: the first `keep` scalar entries (names cut to 40 characters)"""
    out = {}
    for k, v in cfg.items():
        if isinstance(v, (dict, list)):
            continue
        out[k[:40]] = v
        if len(out) >= keep:
            break
    return out


class _StdoutToStderr:
    """RCCL prints a version banner on fd 1 when a communicator is created; the bench contract is ONE JSON line on stdout"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)   # (the banner sits in the C library's buffer when stdout is not a terminal)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the driver's
    command line does (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), and become that launcher."""
    import torch
    have = torch.cuda.device_count()
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.gpus > have and not args.oversubscribe:
        raise SystemExit("bench.py: --gpus %d but this node has %d GPU(s); a run that would print n_gpus=%d without using them is "
                         "refused (development: --oversubscribe puts every rank on device 0)" % (args.gpus, have, args.gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--junctions", type=int, default=10000, help="junctions per GPU per step")
    ap.add_argument("--repeats", type=int, default=25, help="N = 1: how often the K-step region is timed (value = the median region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the U_full / insertion side measurements")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the pipelined host-buffer measurement")
    ap.add_argument("--no-alone", action="store_true", help="skip the one-launch-at-a-time pass behind the timed region (profiling runs)")
    ap.add_argument("--only-extras", default="", help="comma-separated names: run just these side measurements")
    ap.add_argument("--gather", choices=("both", "shm", "rccl"), default="both",
                    help="N > 1: how the results of a step reach rank 0 -- shm: every rank returns its own share into a POSIX shared-memory "
                         "segment rank 0 has mapped, pipelined (dellyhip_batch_fetch_begin / _end; no collective): this is `value` and "
                         "`config.shm_return_*`; rccl: the blocking dellyhip_gather_results to rank 0's HBM (RCCL ncclSend / ncclRecv over xGMI, "
                         "the all-gatherv BASELINE's north_star names) + D2H there: `config.gather_*` (`value` when it is the only path); "
                         "both (default): the two timed regions one after the other in the same run")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="development / the two-process test on a one-GPU box: every rank drives device 0, torch.distributed runs on gloo, "
                         "and the gather protocol runs on the shared-memory transport (dellyhip_comm_create_hostlink) because RCCL "
                         "refuses a communicator whose ranks share a device")
    ap.add_argument("--force-comm", action="store_true",
                    help="development: take the N > 1 code path (MULTI_RESIDENT_BATCHES resident batches, RCCL communicator, gather of step k-1 "
                         "overlapping step k) on ONE GPU with a one-rank communicator")
    ap.add_argument("--dump-rank0-view", default="", help="N > 1: rank 0 writes what it holds after the last step of each return path "
                                                          "(records + blob per path) to this .npz (tests compare it with the checker)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _launch_ranks(args)        # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to print a line whose n_gpus is not "
                         "the number of ranks that ran" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.oversubscribe:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device %d (%d visible); --oversubscribe shares device 0" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _StdoutToStderr():
            dist.init_process_group("gloo" if args.oversubscribe else "nccl", rank=rank, world_size=world)  # nccl == RCCL on ROCm
    dev = torch.device("cpu") if args.oversubscribe else torch.device("cuda:%d" % local)   # where the control-plane tensors live

    from delly_amd import build as dbuild
    from delly_amd import abi, refine, synth
    if rank == 0:
        dbuild.build_lib()
    if world > 1:
        dist.barrier()

    import numpy as np
    n = args.junctions
    multi = world > 1 or args.force_comm
    n_res = MULTI_RESIDENT_BATCHES if multi else RESIDENT_BATCHES
    # weak scaling: rank r owns the junctions [(k * world + r) * n, +n) of the synthetic stream, k = 0 .. n_res - 1
    raw = [synth.make_batch(n, mode="c2", first=(k * world + rank) * n) for k in range(n_res)]
    chroms, batches = one_genome(synth, raw)
    batch = batches[0]
    if multi:
        # the download of step k - 1 (two small compaction kernels in front of it) runs while the persistent sparse kernel of
        # step k holds the chip: 12 of its 16 wavefronts per CU leave them room (as in the slots of dellyhip_stream)
        os.environ.setdefault("DELLYHIP_SPS_WAVES", "12")
    # N > 1: the results travel (to rank 0's host memory, inside every step) with the compact payload -- records + consensus bytes; the
    # "REF,ALT" strings of the deletions are re-cut by the merging process from the record and its own copy of the chromosome
    # (dellyhip_recut_alleles, tests/test_gpu_compact.py): ~0.35 instead of ~1 KB per junction over PCIe / xGMI
    ctx = refine.Context(params=abi.params_sr(compact_alleles=multi or bool(os.environ.get("BENCH_COMPACT"))), device=local)
    ctx.set_chromosomes(chroms)
    # N = 1: consecutive steps alternate between TWO contexts (two scratch areas, one resident genome) on the two compute
    # streams the library verified to run side by side, so the tail of one step's launch runs under the head of the next
    # (one 10 000-junction launch alone: 2.4 wavefronts per resident slot, a 50 us ramp and a 180 us tail of 0.40 ms).
    # That is how the pipelined host path runs them too (dellyhip_stream).  N > 1: the return of step k - 1's results overlaps
    # the kernels of step k.
    # (N > 1 too: the resident batches alternate between two contexts, so the return of step k - 1's results -- compaction kernels
    #  and downloads on that batch's own stream -- is not queued behind the kernels of step k)
    ctxs = [ctx, refine.Context(device=local, share_with=ctx)]
    streams = list(ctx.compute_streams())[:len(ctxs)]
    rbs = [ctxs[k % len(ctxs)].upload(b) for k, b in enumerate(batches)]

    # ---- the return paths of an N > 1 step -------------------------------------------------------------------------------
    paths = []
    if multi:
        paths = ["rccl", "shm"] if args.gather == "both" else [args.gather]
    comm = None
    comm_info = None
    pinned = None
    seg = None
    segs_all = []
    rb_bytes = abi.result_dtype().itemsize
    if "rccl" in paths:
        # every rank alternates between TWO resident batches; the results of the batch refined in the previous step are gathered to
        # rank 0 -- dellyhip_gather_results in the host library: the transport called directly (one all-gather of the (count, bytes)
        # pairs, one of the root's readiness, grouped sends / receives of the records + consensus / allele bytes)
        # -- AND copied into rank 0's pinned host memory (VCF emission needs them there) while the kernels of the current step
        # run on their own stream.  The 128-byte RCCL id travels through torch.distributed.
        with _StdoutToStderr():
            if args.oversubscribe:
                comm = refine.Comm(ctx, rank, world, hostlink="bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), world))
            else:
                ids = [refine.comm_unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(ids, src=0)
                comm = refine.Comm(ctx, rank, world, ids[0])
            if world > 1:
                dist.barrier()   # (torch's own communicator comes up here: its banner too)
        comm_info = comm.info()
        if comm_info["transport_ranks"] != world:
            raise SystemExit("bench.py: the %s transport reports %d ranks, %d were launched" % (comm_info["kind"], comm_info["transport_ranks"], world))
        if rank == 0:
            cap_n = world * n + 64
            cap_b = world * n * 1400 + (1 << 20)
            pinned = (torch.empty(cap_n * rb_bytes, dtype=torch.uint8).pin_memory(), torch.empty(cap_b, dtype=torch.uint8).pin_memory())
    if "shm" in paths:
        # every rank downloads the results of the previous step -- compacted on the device, inside the step, while the kernels of
        # the current step run -- into a POSIX shared-memory segment it owns and has pinned (dellyhip_host_register); rank 0, which
        # would run mergeSort / write the VCF, maps every rank's segment and reads the records in place.  Every rank uses its OWN
        # PCIe link and no collective carries results (CHANGELOG.md 5: the gather funnels all of them through rank 0's link).
        from delly_amd import shmreturn
        tag = "%s_%d" % (os.environ.get("MASTER_PORT", "0"), world)
        cap_n, cap_b = n + 64, n * 1400 + (1 << 20)
        seg = shmreturn.Segment(tag, rank, cap_n, rb_bytes, cap_b, create=True)
        seg.pin(ctx)
        if world > 1:
            dist.barrier()
        if rank == 0:
            segs_all = [seg] + [shmreturn.Segment(tag, r, cap_n, rb_bytes, cap_b, create=False) for r in range(1, world)]
    path_text = {
        "rccl": ("dellyhip_gather_results: %s of records + consensus/allele bytes to rank 0's HBM, then D2H into its pinned host memory, all "
                 "inside the step; gather of step k-1 overlaps the kernels of step k"
                 % ("shared-memory transport (hostlink; RCCL refuses ranks that share a device)" if args.oversubscribe else "RCCL ncclSend/ncclRecv over xGMI")),
        "shm": ("per-rank return (dellyhip_batch_fetch_begin / _end: compaction + a kernel that writes over the rank's own PCIe link) into a pinned "
                "POSIX shared-memory segment mapped by rank 0; step k queues the return of step k-1 behind its kernels and publishes the return "
                "of step k-2; no collective, no host wait on the two youngest steps"),
        None: "none (one GPU: results stay in HBM; host_inclusive has the rate with H2D / D2H)"}
    gathered_n = [0, 0]
    gather_s = [0.0]
    in_flight = [None]
    k_step = [0]
    mode = [paths[0] if paths else None]

    def step():
        i = k_step[0] % len(rbs)
        cur = rbs[i]
        cur.run(streams[i % len(streams)])
        if mode[0] == "rccl" and k_step[0] > 0:
            prev = rbs[(k_step[0] - 1) % len(rbs)]
            tg = time.perf_counter()
            gathered_n[0], gathered_n[1] = prev.gather_into(comm, 0, pinned)   # (waits for prev's kernels, not for cur's)
            gather_s[0] += time.perf_counter() - tg
        elif mode[0] == "shm" and k_step[0] > 0:
            # pipelined: the return of step k - 2 (queued in step k - 1) is waited for and published, then the return of step k - 1 is
            # QUEUED behind its kernels (dellyhip_batch_fetch_begin: compaction + a kernel that writes into the pinned segment) --
            # no host wait on anything younger than two steps, two launches in flight as at N = 1
            prev = rbs[(k_step[0] - 1) % len(rbs)]
            tg = time.perf_counter()
            drain()
            seg.begin()
            prev.fetch_begin(seg.records_view(), seg.blob_view())
            in_flight[0] = prev
            gather_s[0] += time.perf_counter() - tg
        k_step[0] += 1

    def drain():
        """the fetch in flight (if any) has arrived and is published in the segment"""
        if in_flight[0] is not None:
            used = in_flight[0].fetch_end()
            seg.commit(in_flight[0].n, used)
            in_flight[0] = None

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(warm):
        """`warm` untimed steps, then EXACTLY args.steps steps bracketed by barrier + synchronize -> (seconds, gather seconds)"""
        k_step[0] = 0
        for _ in range(warm):
            step()
        drain()
        torch.cuda.synchronize()
        gather_s[0] = 0.0
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        tg = time.perf_counter()
        drain()              # (inside the timed region: the last step's return has reached the segment too)
        gather_s[0] += time.perf_counter() - tg
        sync_all()
        return time.perf_counter() - t0, gather_s[0]

    for i, x in enumerate(rbs):   # set-up, not a step: every resident batch has run once (workspaces sized, results fetchable)
        x.run(streams[i % len(streams)])
    torch.cuda.synchronize()

    def max_over_ranks(sec):
        t = torch.tensor([sec], dtype=torch.float64, device=dev)
        per = [sec]
        if world > 1:
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per = [float(x.item()) for x in allt]
        return max(per), per

    region = {}          # per return path: seconds (max over ranks), per-rank seconds, gather seconds on rank 0, what rank 0 holds
    rank0_view = {}
    if not multi:
        for x in rbs:
            x.kernel_ms()  # reset the kernel timers
        dts = []
        for rep in range(max(1, args.repeats)):
            d, _ = timed_region(args.warmup if rep == 0 else 0)
            dts.append(d)
        order = sorted(dts)
        dt = order[(len(order) - 1) // 2]        # the median region: a region that was actually timed, K steps between two synchronisations
        region[None] = {"dt": dt, "per_rank": [dt], "dts": dts}
    else:
        for pth in paths:
            mode[0] = pth
            for x in rbs:
                x.kernel_ms()
            # the median of a few K-step regions, as at N = 1 (each one: K steps between barrier + synchronize on both sides, max over ranks).
            # A single region of this path has been seen at 0.27 and at 3.1 ms per step on the same build, in processes minutes apart
            # (profiles/r06/README.md: not reproduced by freeing memory, by the order of the rows or by where stdout / stderr go).
            regs = []
            for rep in range(max(1, min(args.repeats, 5))):
                d, gs = timed_region(max(args.warmup, 1) if rep == 0 else 1)
                mx, per = max_over_ranks(d)
                regs.append((mx, per, gs))
            regs.sort(key=lambda r: r[0])
            mx, per, gs = regs[(len(regs) - 1) // 2]
            region[pth] = {"dt": mx, "per_rank": per, "gather_s": gs, "records": gathered_n[0], "blob_bytes": gathered_n[1],
                           "regions_ms_per_step": [r[0] / args.steps * 1e3 for r in regs]}
            if pth == "shm":
                if world > 1:
                    dist.barrier()   # every rank has committed its last download
                if rank == 0:        # what the merging process sees: every rank's last batch, read in place
                    seen, recs, blobs = [], [], []
                    for sg in segs_all:
                        got = sg.read(abi.result_dtype())
                        seen.append(None if got is None else {"rank": sg.rank, "batches_committed": int(got[0]), "records": int(got[1].shape[0]),
                                                              "refined_ok": int(got[1]["ok"].sum()), "blob_bytes": int(got[2].shape[0])})
                        if got is not None and args.dump_rank0_view:
                            recs.append(np.array(got[1])); blobs.append(np.array(got[2]))
                    region[pth]["segments"] = seen
                    region[pth]["records"] = sum(x["records"] for x in seen if x)
                    region[pth]["blob_bytes"] = sum(x["blob_bytes"] for x in seen if x)
                    if args.dump_rank0_view:
                        rank0_view["shm"] = (recs, blobs)
            elif rank == 0 and args.dump_rank0_view:
                nr, nb = gathered_n
                rank0_view["rccl"] = (np.frombuffer(pinned[0].numpy()[:nr * rb_bytes].tobytes(), dtype=abi.result_dtype()), pinned[1].numpy()[:nb].copy())
    # N > 1: `value` is the pipelined per-rank return (shm) when it was timed; the blocking gather is timed beside it (config.gather_*)
    first = ("shm" if "shm" in paths else paths[0]) if paths else None
    dt = region[first]["dt"]
    per_rank_ms = [x / args.steps * 1e3 for x in region[first]["per_rank"]]
    kms = [x.kernel_ms() for x in rbs]          # (sync + averages over each batch's launches; N > 1: of the last return path)
    dps = [x.dp_kernel_ms() for x in rbs]
    used = [i for i, k in enumerate(kms) if k[2] > 0]
    launches = sum(kms[i][2] for i in used)
    ms_split = sum(kms[i][0] * kms[i][2] for i in used) / max(launches, 1)
    ms_dp = sum(dps[i] * kms[i][2] for i in used) / max(launches, 1)
    ran = torch.tensor([1 if launches > 0 else 0], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(ran)
    ranks_that_ran = int(ran.item())

    # sanity: the timed work is the real work (every junction refined; tests/test_gpu_bench_shapes.py compares exactly these
    # batches with oracle/_ref)
    n_ok = [int(x.fetch()[0]["ok"].sum()) for x in rbs]
    if rank == 0 and args.dump_rank0_view:
        dump = {"world": np.int64(world), "junctions": np.int64(n), "steps": np.int64(args.steps), "warmup": np.int64(max(args.warmup, 1))}
        if "rccl" in rank0_view:
            dump["rccl_records"], dump["rccl_blob"] = rank0_view["rccl"]
        for r, (rec, bl) in enumerate(zip(*rank0_view.get("shm", ([], [])))):
            dump["shm_records_%d" % r], dump["shm_blob_%d" % r] = rec, bl
        np.savez(args.dump_rank0_view, **dump)

    # the same launches ONE AT A TIME (rounds 1-2's headline mode): what a launch costs when it has the chip to itself
    alone = None
    if not multi and not args.no_alone:
        for x in rbs:
            x.kernel_ms()
        reps = 24
        ta = time.perf_counter()
        for k in range(reps):
            x = rbs[k % len(rbs)]
            x.run(streams[0])
            x.sync()
        da = (time.perf_counter() - ta) / reps
        kk = [x.kernel_ms() for x in rbs]
        dd = [x.dp_kernel_ms() for x in rbs]
        la = sum(k[2] for k in kk)
        alone = {"alignments_per_s": n / da, "ms_per_step": da * 1e3,
                 "kernel_ms": sum(d * k[2] for d, k in zip(dd, kk)) / max(la, 1), "launches": la,
                 "note": "each launch waits for the previous one (sync in between)"}
        alone["achieved"] = n * ALG_BYTES_PER_U / (alone["kernel_ms"] * 1e-3) / 1e9 if alone["kernel_ms"] > 0 else 0.0
        alone["frac"] = alone["achieved"] / HBM_PEAK_GBS

    # the same junctions from host buffers to host buffers through the pipelined path (every rank its own share)
    hi = None
    try:
        if not args.no_host_inclusive:
            for x in rbs:
                x.free()
            rbs = []
            # The caller-facing configuration (include/delly_dropin/split.h: refineBatch): compact payload -- the "REF,ALT" strings of
            # small deletions (~700 of the ~1000 result bytes per junction) are plain substrings of the chromosome the caller holds and
            # are re-cut there from the record (dellyhip_recut_alleles; tests/test_gpu_compact.py holds them to oracle/_ref), so they
            # do not cross PCIe.  The full payload (round 5's figure) is timed beside it, and so is the host re-cut of one batch.
            ctx_c = refine.Context(params=abi.params_sr(compact_alleles=True), device=local, share_with=ctx)
            hi = host_inclusive_rate(ctx_c, batches, 0, pinned_input=True)
            hi["input"] = "sequence bytes in pinned host memory, read in place (dellyhip_stream_zero_copy); records and offsets staged"
            hi["payload"] = "compact: records + consensus bytes; REF,ALT re-cut on the host from the record (DELLYHIP_COMPACT_ALLELES)"
            try:
                if world == 1:
                    full = host_inclusive_rate(ctx, batches, 0, seconds=0.5)
                    hi["full_payload"] = {k: full[k] for k in ("value", "ms_per_batch", "bytes_down_per_batch", "batches")}
                    hi["full_payload"]["input"] = "pageable host buffers, staged by submit (round 5's configuration)"
                    mid = host_inclusive_rate(ctx_c, batches, 0, seconds=0.5)
                    hi["compact_payload_pageable_input"] = {k: mid[k] for k in ("value", "ms_per_batch", "bytes_down_per_batch", "batches")}
                    hi["two_caller_threads"] = host_inclusive_rate_threads(ctx_c, batches, 0, 2, seconds=0.7)
                    gr, gb = ctx_c.refine(batches[0], want_alignment=False)
                    best = None
                    buf = None
                    for _ in range(5):
                        buf, _, sec = refine.recut_alleles_raw(ctx_c.params, batches[0].junctions, gr, gb, chroms, out=buf)
                        best = sec if best is None else min(best, sec)
                    hi["host_recut_ms_per_batch_one_thread"] = best * 1e3
                    hi["host_recut_bytes_per_batch"] = int(buf.nbytes)
                    hi["host_recut_note"] = ("dellyhip_recut_alleles_batch on ONE host thread over one batch's records: the VCF writer's job, off the refinement "
                                             "stage's clock (src/split.h:606-624 fills sv.alleles for output only); not inside `value`")
            except Exception as e:
                hi["full_payload"] = {"error": repr(e)}
            ctx_c.close()
            if world > 1:
                hv = torch.tensor([hi["value"], hi["wall_s"]], dtype=torch.float64, device=dev)
                allh = [torch.zeros_like(hv) for _ in range(world)]
                dist.all_gather(allh, hv)
                hi["per_rank_junctions_per_s"] = [float(x[0].item()) for x in allh]
                hi["value"] = float(sum(x[0].item() for x in allh))   # independent shards, no exchange in this leg
                hi["note"] += "; N > 1: sum over ranks, every rank streams its own shard (results stay on each rank's host)"
    except Exception as e:  # the contract line must still come out
        hi = {"error": repr(e)}

    if rank == 0:
        total_units = world * n * args.steps
        value = total_units / dt
        ach = n * ALG_BYTES_PER_U / (ms_dp * 1e-3) / 1e9 if ms_dp > 0 else 0.0
        tr = _measured_traffic()
        traffic = tr["traffic"]
        # the roof that binds (integer VALU issue): wave64 VALU instructions of one launch (SQ_INSTS_VALU, same committed
        # PMC file as `traffic`) / (SIMDs x measured clock / 2 cycles per instruction) / the kernel's time in THIS run
        valu_roof = 1024 * VALU_CEILING["sclk_mhz_measured"] * 1e6 / VALU_CEILING["cycles_per_wave64_valu_at_8_waves_per_simd"]
        valu_frac = (tr["valu"] / valu_roof / (ms_dp * 1e-3)) if (tr["valu"] and ms_dp > 0) else None
        cfg = {"workload": "BASELINE configs[1]: %d synthetic DEL junctions per GPU and step, 150 bp consensus x 1 kb ref window" % n,
               "workload_detail": "alignConsensus (longNeedle + split detection), bit-exact; steps rotate through %d different resident batches%s"
                                  % (len(batches), "" if multi else ", consecutive steps on two contexts / two HIP streams (two launches in flight)"),
               "launches_in_flight": 1 if (multi and first != "shm") else 2,
               "junctions_per_gpu": n, "resident_batches": len(batches), "refined_ok_min": min(n_ok), "parallelism": "junction-sharded x%d" % world,
               "ranks_launched": world, "ranks_that_ran_kernels": ranks_that_ran,
               "value_is": ("inputs resident in HBM, results left in HBM (the bench contract); host buffers in -> host buffers out is host_inclusive_alignments_per_s "
                            "(compact payload) / host_inclusive_full_payload_alignments_per_s") if not multi else
                           ("inputs resident in HBM; the previous step's results (compact payload) reach rank 0's host memory inside every step through the return path "
                            "config.value_return_path (shm = per-rank PCIe writes into shared memory, no collective; the RCCL all-gatherv to rank 0 is gather_alignments_per_s)"),
               "kernels_ms_per_step_rank0": ms_split}
        if not multi:
            dts = region[None]["dts"]
            rates = sorted(world * n * args.steps / d for d in dts)
            cfg.update({"timed_regions": len(dts), "value_is_region": "median of the timed K-step regions",
                        "value_min": rates[0], "value_max": rates[-1], "value_median": rates[(len(rates) - 1) // 2],
                        "timed_seconds_total": sum(dts)})
            if alone:   # flat copies: the driver's record keeps scalars of `config` / `roofline` only
                cfg.update({"one_launch_at_a_time_alignments_per_s": alone["alignments_per_s"], "one_launch_at_a_time_ms_per_step": alone["ms_per_step"],
                            "one_launch_at_a_time_kernel_ms": alone["kernel_ms"]})
        else:
            cfg["return_path"] = path_text[first]
            cfg["value_return_path"] = first
            cfg["ms_per_step_min_rank"] = min(per_rank_ms)
            cfg["ms_per_step_max_rank"] = max(per_rank_ms)
            if comm_info is not None:
                cfg["rccl_ranks"] = comm_info["transport_ranks"]     # ncclCommCount (hostlink: attached processes)
                cfg["gather_transport"] = comm_info["kind"]
            cfg["oversubscribed_one_device"] = bool(args.oversubscribe)
            if "rccl" in region:   # the blocking gather to rank 0 (dellyhip_gather_results), its own timed region
                r = region["rccl"]
                cfg["gather_alignments_per_s"] = total_units / r["dt"]
                cfg["gather_step_ms"] = r["dt"] / args.steps * 1e3
                cfg["gather_ms_per_step"] = r["gather_s"] / max(args.steps, 1) * 1e3
                cfg["gathered_records_on_rank0"] = r["records"]
                cfg["gathered_blob_bytes_on_rank0"] = r["blob_bytes"]
                cfg["gather_path"] = path_text["rccl"]
                cfg["gather_regions_ms_per_step"] = r.get("regions_ms_per_step")
            if "shm" in region:    # the pipelined per-rank return into shared memory
                r = region["shm"]
                cfg["shm_return_alignments_per_s"] = total_units / r["dt"]
                cfg["shm_return_ms_per_step"] = r["dt"] / args.steps * 1e3
                cfg["shm_return_gather_ms_per_step"] = r["gather_s"] / max(args.steps, 1) * 1e3
                cfg["shm_return_records_seen_by_rank0"] = r["records"]
                cfg["shm_return_blob_bytes_seen_by_rank0"] = r["blob_bytes"]
                cfg["shm_return_path"] = path_text["shm"]
                cfg["shm_return_regions_ms_per_step"] = r.get("regions_ms_per_step")
                cfg["value_is_region"] = "median of %d timed K-step regions (max over ranks each)" % len(r.get("regions_ms_per_step") or [0])
        if isinstance(hi, dict) and "value" in hi:
            cfg["host_inclusive_alignments_per_s"] = hi["value"]       # SURVEY.md 8d's definition: host buffers in -> host buffers out
            cfg["host_inclusive_wall_s"] = hi["wall_s"]
            cfg["host_inclusive_batches"] = hi["batches"]
            cfg["host_inclusive_ms_per_batch"] = hi["ms_per_batch"]
            cfg["host_inclusive_payload"] = "compact (REF,ALT re-cut on the host from the record); full payload: host_inclusive_full_payload_alignments_per_s"
            if isinstance(hi.get("full_payload"), dict) and "value" in hi["full_payload"]:
                cfg["host_inclusive_full_payload_alignments_per_s"] = hi["full_payload"]["value"]
            if isinstance(hi.get("two_caller_threads"), dict) and "value" in hi["two_caller_threads"]:
                cfg["host_inclusive_two_caller_threads_per_s"] = hi["two_caller_threads"]["value"]
            if "host_recut_ms_per_batch_one_thread" in hi:
                cfg["host_recut_ms_per_batch_one_thread"] = hi["host_recut_ms_per_batch_one_thread"]
        out = {
            "metric": "candidate split-read alignments/sec (DEL, 150bp reads, 1kb ref window)",
            "value": value,
            "unit": "alignments/s",
            "n_gpus": ranks_that_ran if not args.oversubscribe else 1,   # distinct devices that ran kernels (every oversubscribed rank drives device 0; config.ranks_launched has the rank count)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32-exact (uint8 / int16 storage)",
            "data": "synthetic",
            "config": cfg,
            "host_inclusive": hi,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": tr["stale"], "traffic_source": tr["source"],
                         "valu_frac": valu_frac, "valu_wave_instructions_per_launch": tr["valu"], "valu_roof_wave_instr_per_s": valu_roof,
                         "valu_frac_alone": (tr["valu"] / valu_roof / (alone["kernel_ms"] * 1e-3)) if (tr["valu"] and alone and alone["kernel_ms"] > 0) else None,
                         "kernel": "split_sparse_kernel (sparse longNeedle, one junction per wavefront, alignment + split detection fused)",
                         "kernel_ms": ms_dp, "all_split_kernels_ms": ms_split, "kernel_launches_timed": launches,
                         "kernel_ms_is": ("HIP events around each launch; two launches in flight share the chip (kernel_ms_alone: isolated)"
                                          if not multi else "HIP events around each launch on its stream over the timed region"),
                         "kernel_ms_alone": alone["kernel_ms"] if alone else None,
                         "achieved_alone": alone["achieved"] if alone else None,
                         "frac_alone": alone["frac"] if alone else None,
                         "one_launch_at_a_time": alone,
                         "alg_bytes_per_launch": n * ALG_BYTES_PER_U,
                         "binding_roof": "integer VALU issue / latency, DP state on chip (SURVEY.md 8d); the HBM fraction is reported because BASELINE asks for it",
                         "valu_cycles_per_wave64_instr_at_8_waves_per_simd": VALU_CEILING["cycles_per_wave64_valu_at_8_waves_per_simd"],
                         "valu_cycles_per_instr_one_wave": VALU_CEILING["cycles_per_instruction_one_wave"],
                         "sclk_mhz_measured": VALU_CEILING["sclk_mhz_measured"],
                         "valu_ceiling": VALU_CEILING,
                         "note": "the kernel's cost depends on the junctions' deficits (extras.deficit_sweep); no GCUPS figure"},
        }
        if multi:
            out["config"]["segments_seen_by_rank0"] = region.get("shm", {}).get("segments")
            out["config"]["ms_per_step_per_rank"] = per_rank_ms
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras:
            only = set(filter(None, args.only_extras.split(','))) or None
            out["extras"] = side_measurements(ctx, synth, device=local, with_cpu=not args.no_cpu_baseline, only=only)
            if only is None or "deficit_sweep" in only:
                try:
                    out["extras"]["deficit_sweep"] = deficit_sweep(refine.Context(device=local), synth, local, with_cpu=not args.no_cpu_baseline)
                except Exception as e:  # side figure only
                    out["extras"]["deficit_sweep"] = {"error": repr(e)}
            # flat copies of the side rows the verdicts track: the driver's record keeps the FIRST 24 scalars of `config`
            # (BENCH_r04.json lost every side row to five descriptive strings in front of them), see order_config()
            for name, key in FLAT_ROWS:
                row = out["extras"].get(name)
                if isinstance(row, dict) and "junctions_per_s" in row:
                    out["config"][key] = row["junctions_per_s"]
            row = out["extras"].get("u_full_n20_10k_junctions")
            if isinstance(row, dict) and "msa_deferred_junctions" in row:
                out["config"]["u_full_n20_10k_msa_deferred_junctions"] = row["msa_deferred_junctions"]
            if isinstance(row, dict) and isinstance(row.get("host_inclusive"), dict) and "value" in row["host_inclusive"]:
                # the same batch from host buffers to host buffers through dellyhip_stream: several launches in flight, the tail of one under
                # the head of the next (the resident row above runs one launch at a time)
                out["config"]["u_full_n20_10k_host_inclusive_per_s"] = row["host_inclusive"]["value"]
            sw = out["extras"].get("deficit_sweep")
            if isinstance(sw, dict):
                for name, row in sw.items():
                    if isinstance(row, dict) and "alignments_per_s" in row:
                        out["config"]["deficit_sweep_%s_alignments_per_s" % name] = row["alignments_per_s"]
                for name in ("substitutions_2pct", "substitutions_5pct"):   # (short names: the driver cuts keys at 40 characters)
                    if isinstance(sw.get(name), dict) and "alignments_per_s" in sw[name]:
                        out["config"]["%s_alignments_per_s" % name] = sw[name]["alignments_per_s"]
        out["config"] = order_config(out["config"])
        print(json.dumps(out), flush=True)
    for x in rbs:
        x.free()
    if comm is not None:
        comm.close()
    for sg in segs_all[1:]:
        sg.close()
    if world > 1 and seg is not None:
        dist.barrier()       # (readers unmap before the owners unlink)
    if seg is not None:
        seg.close()
    for cx in ctxs[1:]:
        cx.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
