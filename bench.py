#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: candidate split-read alignments/sec
(DEL, 150 bp consensus, 1 kb reference window) on N MI355X.

A "step" is one pass of the hot path (alignConsensus: window construction,
longNeedle, split detection, coordinates) over one resident batch of synthetic
junctions (BASELINE config 2: 10 000 junctions per GPU).  Inputs
are in HBM before the timed region; result records stay in HBM and, for N > 1,
are gathered to every rank with one RCCL all_gather per step (junctions shard
across ranks, no other exchange).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task description) with the
`roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_U = 1214        # SURVEY.md 8d / BASELINE.md 4: m + n + 64 B record at C2
CELLS_PER_U = 2 * 151 * 1001  # fwd + rev DP cells
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


# The side measurements of side_measurements(): (name, junctions, CPU sample, synth.make_batch kwargs).
# tests/test_gpu_bench_shapes.py bit-compares the HIP path with oracle/_ref on exactly these batches.
SIDE_PLAN = (("u_c2_40k_junctions", 40000, 0, dict(mode="c2")),
             ("u_full_n20", 2000, 2000, dict(mode="c2", n_reads=20)),
             ("u_full_n20_10k_junctions", 10000, 0, dict(mode="c2", n_reads=20)),   # the chip filled: one wavefront per junction needs > 4 096 of them
             ("u_full_n5", 2000, 2000, dict(mode="c2", n_reads=5)),
             ("ins_svt4", 5000, 5000, dict(mode="ins")),
             ("lr_c4_align_consensus", 2048, 128, dict(mode="lr", sub_rate=0.01)),
             ("lr_c4_msaedlib_n15", 768, 64, dict(mode="lr", n_reads=15, sub_rate=0.06)),
             # INS 800 bp, 15 reads of ~3.8 kb at 6 % error: msaWfa + alignConsensus (splitAlign)
             ("lr_ins_msawfa_n15", 512, 64, dict(mode="lrins", n_reads=15, sub_rate=0.06)))


class _DevPtr:
    """Wraps a raw device pointer for torch.as_tensor (RCCL needs a tensor)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def cpu_baseline(batch, budget_s=12.0):
    """The reference's CPU path (oracle/_ref: its own headers) or the C port, timed on this box's host cores with
    the reference's threading model (src/shortpe.h:175-201: std::threads on one atomic counter) over the SAME
    10 000 C2 junctions.  Only the loop body is timed -- ONE alignConsensus() per junction, no diagnostic replay,
    no marshalling -- and the clock runs inside the C++ driver around thread start .. join
    (oracle/ref_driver.cpp: dref_time_refine_batch).  Every thread gets >= 32 junctions per pass."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    orc = pyoracle.Oracle(kind)
    cores = os.cpu_count() or 1
    # single thread: ~1 s on a 512-junction prefix
    cal = _subbatch(batch, 512)
    s1, n1, _ = orc.time_refine(cal, n_threads=1, reps=1)
    rate1 = n1 / s1
    # all cores: the whole batch (>= 32 junctions per thread, else fewer threads), repeated to ~budget_s
    threads = max(1, min(cores, batch.n // 32))
    s0, _, _ = orc.time_refine(batch, n_threads=threads, reps=1)   # (also warms the thread stacks / page cache)
    reps = int(max(1, min(400, budget_s / max(s0, 1e-3))))
    sN, nN, okN = orc.time_refine(batch, n_threads=threads, reps=reps)
    return {"value": nN / sN, "unit": "alignments/s", "cores": threads, "kind": kind,
            "value_one_thread": rate1,
            "sample": "%d passes over the same %d C2 junctions (%d alignConsensus calls, %d returned true) on %d "
                      "std::thread workers pulling from one atomic counter (src/shortpe.h:175-201 model), %.1f s "
                      "measured inside the C++ driver around thread start..join; one thread: %d junctions in %.2f s"
                      % (reps, batch.n, nN, okN, threads, sN, n1, s1)}


def _profile_file(name):
    """newest committed copy of a profile artefact (profiles/rNN/<name>)"""
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            return path
    return None


def _measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); None if not collected."""
    try:
        with open(_profile_file("pmc_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        return None


# VALU issue ceiling of one SIMD for the op mix of the DP kernels (v_pk_*_i16, v_max, DPP moves, v_bfi ...), measured with
# tools/valu_rate.hip at 4-8 resident wavefronts per SIMD, >= 50 ms kernels: profiles/r02/valu_rate.txt (0.51-0.55 G
# wave-instructions/s; plain v_add / v_and reach 0.73 G).  Nominal figure of MI355X_MICROARCH.md: a wave64 VALU
# instruction issues over 2 cycles = 1.2 G/s per SIMD at 2.4 GHz (157.3 TFLOP/s fp32).
VALU_PEAK_MEASURED_PER_SIMD = 0.55e9
VALU_PEAK_NOMINAL_PER_SIMD = 1.2e9
N_SIMD = 1024


def _valu_instructions(kernel_prefix, n_junctions):
    """VALU wave-instructions per launch of the dominant kernel from the committed SQ counter pass
    (profiles/rNN/pmc_sq_summary.txt, SQ_INSTS_VALU, collected at 10 000 C2 junctions), scaled to this launch"""
    try:
        for line in open(_profile_file("pmc_sq_summary.txt")):
            if kernel_prefix in line and "SQ_INSTS_VALU" in line:
                import ast
                d = ast.literal_eval(line[line.index("{"):])
                return d["SQ_INSTS_VALU"] * (n_junctions / 10000.0)
    except Exception:
        pass
    return None


def _subbatch(batch, n):
    from delly_amd import synth
    n = min(n, batch.n)
    first = int(batch.junctions["seq_first"][0])
    last = int(batch.junctions["seq_first"][n - 1] + batch.junctions["n_seq"][n - 1])
    return synth.Batch(batch.chroms, batch.junctions[:n].copy(), batch.seq_blob, batch.seq_off[:last + 1].copy(),
                       batch.with_msa, batch.truth[:n])


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
    # SURVEY.md 8d "GPU time includes H2D/D2H and host marshalling": the same 10 000 C2 junctions through the host-buffer
    # entry point dellyhip_align_consensus_batch -- upload of records + consensus bytes, host binning, kernels,
    # device-side compaction, download of records + consensus / allele bytes (chromosome resident).  Never `value`.
    try:
        if not want("u_c2_host_inclusive"):
            raise KeyError("skipped")
        bb = synth.make_batch(10000, mode="c2")
        ctx.set_chromosomes(bb.chroms)
        ctx.refine(bb)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            res, blob = ctx.refine(bb)
        dt = (time.perf_counter() - t0) / reps
        out["u_c2_host_inclusive"] = {"junctions": bb.n, "junctions_per_s": bb.n / dt, "ms_per_call": dt * 1e3,
                                      "refined_ok": int(res["ok"].sum()), "bytes_up": int(bb.seq_blob.size + bb.junctions.nbytes + bb.seq_off.nbytes),
                                      "bytes_down": int(res.nbytes + blob.size),
                                      "note": "dellyhip_align_consensus_batch from host buffers: H2D + binning + kernels + compaction + D2H"}
    except KeyError:
        pass
    except Exception as e:  # side figure only
        out["u_c2_host_inclusive"] = {"error": repr(e)}
    # the headline batch size with two batches in flight (two contexts = two scratch areas, two HIP streams): one
    # 10 000-junction step is 2500 DP wavefronts, fewer than three per SIMD; overlapping consecutive steps fills the chip
    try:
        if not want("u_c2_two_batches_in_flight"):
            raise KeyError("skipped")
        import torch
        ctx2 = refine.Context(device=device)
        pairs = []
        for cx, first in ((ctx, 0), (ctx2, 10000)):
            bb = synth.make_batch(10000, mode="c2", first=first)
            cx.set_chromosomes(bb.chroms)
            pairs.append((cx.upload(bb), torch.cuda.Stream(device=device)))
        for _ in range(2):
            for rb, st in pairs:
                rb.run(st.cuda_stream)
        for rb, _ in pairs:
            rb.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            for rb, st in pairs:
                rb.run(st.cuda_stream)
        for rb, _ in pairs:
            rb.sync()
        dt = (time.perf_counter() - t0) / (2 * reps)
        ok = sum(int(rb.fetch()[0]["ok"].sum()) for rb, _ in pairs)
        out["u_c2_two_batches_in_flight"] = {"junctions": 10000, "junctions_per_s": 10000 / dt, "ms_per_step": dt * 1e3,
                                             "refined_ok": ok,
                                             "note": "two contexts, each with a resident 10 000-junction batch, alternating on two HIP streams"}
        for rb, _ in pairs:
            rb.free()
        ctx2.close()
    except KeyError:
        pass
    except Exception as e:  # side figure only
        out["u_c2_two_batches_in_flight"] = {"error": repr(e)}
    for name, n, ncpu, kw in plan:
        b = synth.make_batch(n, **kw)
        lr = kw["mode"].startswith("lr")
        params = abi.params_lr(realign=True) if lr else abi.params_sr()
        if lr and name == "lr_c4_align_consensus":  # long-read parameters + orientation test (src/tegua.h:237-241, src/assemble.h:849)
            ctx = refine.Context(params=params, device=device)
        ctx.set_chromosomes(b.chroms)
        rb = ctx.upload(b)
        rb.run(); rb.sync(); rb.kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            rb.run()
        rb.sync()
        dt = (time.perf_counter() - t0) / steps
        ms_split, ms_msa, _ = rb.kernel_ms()
        res, _ = rb.fetch()
        out[name] = {"junctions": n, "junctions_per_s": n / dt, "ms_per_step": dt * 1e3, "msa_stage_ms": ms_msa,
                     "split_stage_ms": ms_split, "refined_ok": int(res["ok"].sum())}
        rb.free()
        if orc is not None and ncpu > 0:
            sub = b if ncpu >= n else synth.make_batch(ncpu, **kw)
            threads = max(1, min(cores, sub.n))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--junctions", type=int, default=10000, help="junctions per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the U_full / insertion side measurements")
    ap.add_argument("--only-extras", default="", help="comma-separated names: run just these side measurements")
    ap.add_argument("--force-comm", action="store_true",
                    help="development: take the N > 1 code path (two resident batches, RCCL communicator, gather of step k-1 "
                         "overlapping step k) on ONE GPU with a one-rank communicator")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # nccl == RCCL on ROCm

    from delly_amd import build as dbuild
    from delly_amd import abi, refine, synth
    if rank == 0:
        dbuild.build_lib()
    if world > 1:
        dist.barrier()

    n = args.junctions
    batch = synth.make_batch(n, mode="c2", first=rank * n)  # weak scaling: shard by junction index
    ctx = refine.Context(device=local)
    side = torch.cuda.Stream(device=local)  # the kernels are launched on this stream
    stream = side.cuda_stream
    comm = None
    gather_kind = "none (one GPU: results stay in HBM)"
    rbs = []
    multi = world > 1 or args.force_comm
    if not multi:
        ctx.set_chromosomes(batch.chroms)
        rbs.append(ctx.upload(batch))
    else:
        # N > 1: every rank keeps TWO resident batches of n junctions and alternates; the results of the batch refined in
        # the previous step are gathered to rank 0's HBM -- dellyhip_gather_results_device in the host library: RCCL called
        # directly (ncclAllGather of the counts, grouped ncclSend / ncclRecv of the records + consensus / allele bytes,
        # SURVEY.md 8e) -- while the kernels of the current step run on their own stream.  One run + one gather per step.
        # The 128-byte RCCL id travels through torch.distributed.
        b2 = synth.make_batch(n, mode="c2", first=(world + rank) * n)
        chrom = __import__("numpy").concatenate([batch.chroms[0], b2.chroms[0]])
        j2 = b2.junctions.copy()
        j2["sv_start"] += batch.chroms[0].size
        j2["sv_end"] += batch.chroms[0].size
        b2 = synth.Batch([chrom], j2, b2.seq_blob, b2.seq_off, b2.with_msa, b2.truth)
        ctx.set_chromosomes([chrom])
        rbs = [ctx.upload(batch), ctx.upload(b2)]
        ids = [refine.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        comm = refine.Comm(ctx, rank, world, ids[0])
        gather_kind = ("dellyhip_gather_results_device: RCCL ncclSend/ncclRecv of records + consensus/allele bytes to rank 0, "
                       "gather of step k-1 overlapping the kernels of step k")
    rb = rbs[0]
    gathered_n = [0, 0]
    k_step = [0]

    def step():
        cur = rbs[k_step[0] % len(rbs)]
        with torch.cuda.stream(side):
            cur.run(stream)
        if comm is not None:
            prev = rbs[(k_step[0] + 1) % 2]
            if k_step[0] > 0:
                gathered_n[0], gathered_n[1] = prev.gather_device(comm, 0)   # (waits for prev's kernels, not for cur's)
        k_step[0] += 1

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for x in rbs:
        x.kernel_ms()  # reset the kernel timers
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_split, ms_msa, launches = rb.kernel_ms()
    ms_dp = rb.dp_kernel_ms()
    for other in rbs[1:]:
        other.sync()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local)
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # sanity: the timed work is the real work (every junction refined, parity spot check vs oracle in smoke())
    res, _ = rb.fetch()
    n_ok = int(res["ok"].sum())

    if rank == 0:
        total_units = world * n * args.steps
        value = total_units / dt
        ach = n * ALG_BYTES_PER_U / (ms_dp * 1e-3) / 1e9 if ms_dp > 0 else 0.0
        traffic = _measured_traffic()
        valu_n = _valu_instructions("split_sparse_kernel", n)
        valu = None
        if valu_n and ms_dp > 0:
            rate = valu_n / (ms_dp * 1e-3)
            valu = {"wave_instructions_per_launch": valu_n, "achieved_G_per_s": rate / 1e9,
                    "peak_measured_G_per_s": VALU_PEAK_MEASURED_PER_SIMD * N_SIMD / 1e9,
                    "frac_of_measured_peak": rate / (VALU_PEAK_MEASURED_PER_SIMD * N_SIMD),
                    "peak_nominal_G_per_s": VALU_PEAK_NOMINAL_PER_SIMD * N_SIMD / 1e9,
                    "frac_of_nominal_peak": rate / (VALU_PEAK_NOMINAL_PER_SIMD * N_SIMD),
                    "source": "SQ_INSTS_VALU of profiles/*/pmc_sq_summary.txt / live kernel time; peaks: profiles/r02/valu_rate.txt "
                              "(tools/valu_rate.hip) and MI355X_MICROARCH.md (wave64 VALU over 2 cycles)"}
        out = {
            "metric": "candidate split-read alignments/sec (DEL, 150bp reads, 1kb ref window)",
            "value": value,
            "unit": "alignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic DEL junctions per GPU, 150 bp consensus x 1 kb "
                                   "ref window, alignConsensus (longNeedle + split detection), bit-exact" % n,
                       "junctions_per_gpu": n, "refined_ok": n_ok, "parallelism": "junction-sharded x%d" % world,
                       "gather": gather_kind, "gathered_per_step_on_rank0": ({"records": gathered_n[0], "blob_bytes": gathered_n[1]} if comm is not None else None),
                       "ms_per_step_per_rank": per_rank_ms, "kernels_ms_per_step_rank0": ms_split},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "split_sparse_kernel (sparse longNeedle: furthest-reaching tables per deficit level, one junction per wavefront, alignment + split detection fused)",
                         "kernel_ms": ms_dp, "all_split_kernels_ms": ms_split,
                         "alg_bytes_per_launch": n * ALG_BYTES_PER_U,
                         "gcups_dense_equivalent": n * CELLS_PER_U / (ms_dp * 1e-3) / 1e9 if ms_dp > 0 else 0.0,
                         "valu": valu, "valu_frac": valu["frac_of_measured_peak"] if valu else None,
                         "note": "path is integer-VALU bound with DP state on chip; HBM fraction is reported because "
                                 "BASELINE asks for it (SURVEY.md 8d)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras:
            rb.free()
            rb = None
            rbs = []
            out["extras"] = side_measurements(ctx, synth, device=local, with_cpu=not args.no_cpu_baseline,
                                                only=set(filter(None, args.only_extras.split(','))) or None)
        print(json.dumps(out), flush=True)
    for x in rbs:
        x.free()
    if comm is not None:
        comm.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
