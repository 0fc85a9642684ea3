"""Host-inclusive throughput of the pipelined path (dellyhip_stream): `--batches` batches of `--n` junctions cycle through
`--distinct` different synthetic batches (one chromosome table), `--depth` slots in flight; the clock covers validation,
routing, staging copies, H2D, kernels, compaction, D2H and the final wait -- everything between host buffers in and host
buffers out.  python tools/bench_stream.py [--mode c2] [--n-reads 0] ..."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import abi, refine, synth  # noqa: E402


def one_genome(batches):
    chroms = [np.concatenate([b.chroms[c] for b in batches]) for c in range(len(batches[0].chroms))]
    out, base = [], [0] * len(chroms)
    for b in batches:
        j = b.junctions.copy()
        j["sv_start"] += base[0]
        j["sv_end"] += np.where(j["chr2"] == 0, base[0], base[-1])
        out.append(synth.Batch(chroms, j, b.seq_blob, b.seq_off, b.with_msa, b.truth))
        base = [x + c.size for x, c in zip(base, b.chroms)]
    return chroms, out


KEPT = []


def stream_rate(ctx, batches, with_msa, depth, total, want_alignment=False):
    """-> dict: junctions/s over `total` submitted batches (cycling through `batches`), host-inclusive"""
    st = refine.Stream(ctx, depth=depth, with_msa=with_msa, want_alignment=want_alignment)
    args = []
    for b in batches:
        junc = np.ascontiguousarray(b.junctions)
        blob = np.ascontiguousarray(b.seq_blob, dtype=np.uint8)
        off = np.ascontiguousarray(b.seq_off, dtype=np.uint64)
        args.append((junc.shape[0], junc.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_char_p),
                     off.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(off.size - 1), (junc, blob, off)))

    def run(count):
        nxt = done_j = done_b = 0
        for k in range(count):
            # depth - 1 batches in flight while the consumer still reads the block of the last collect (depth 1: it has
            # copied what it needs and gives the block back)
            if depth == 1:
                st.release()
            while nxt < count and nxt - k < max(1, depth - 1):
                a = args[nxt % len(args)]
                st.submit_raw(a[0], a[1], a[2], a[3], a[4], nxt)
                nxt += 1
            n, ln = st.collect_raw()
            done_j += n
            done_b += ln
        return done_j, done_b

    run(max(2 * depth, len(args)))          # warm-up: buffers grow to the batch size
    st.stats(reset=True)
    t0 = time.perf_counter()
    nj, nb = run(total)
    dt = time.perf_counter() - t0
    stats = st.stats()
    if os.environ.get("BENCH_STREAM_KEEP"):
        KEPT.append(st)
    else:
        st.close()
    up = sum(a[5][0].nbytes + a[5][1].nbytes + a[5][2].nbytes for a in args) / len(args)
    return {"junctions_per_s": nj / dt, "batches": total, "junctions_per_batch": nj / total, "wall_s": dt, "ms_per_batch": dt / total * 1e3,
            "host_ms_per_batch": {k: (v / total * 1e3 if k.endswith("_s") else v) for k, v in stats.items()},
            "depth": depth, "bytes_up_per_batch": int(up), "bytes_down_per_batch": int(nb / total + nj / total * abi.result_dtype().itemsize)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--batches", type=int, default=100)
    ap.add_argument("--distinct", type=int, default=6)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--mode", default="c2")
    ap.add_argument("--n-reads", type=int, default=0)
    ap.add_argument("--sub-rate", type=float, default=0.005)
    ap.add_argument("--only-depth", type=int, default=0)
    ap.add_argument("--torch", type=int, default=0, help="1: initialise torch.cuda first; 2: and create a torch stream; 3: and resident batches like bench.py")
    ap.add_argument("--depths", default="", help="comma-separated list of depths instead of 1,2,3,--depth")
    args = ap.parse_args()
    keep = []
    if args.torch >= 1:
        import torch
        torch.cuda.init()
        torch.zeros(16, device="cuda:0")
        if args.torch >= 2:
            keep.append(torch.cuda.Stream(device="cuda:0"))
            with torch.cuda.stream(keep[0]):
                torch.zeros(16, device="cuda:0")
            torch.cuda.synchronize()
    lr = args.mode.startswith("lr")
    params = abi.params_lr(realign=True) if lr else abi.params_sr()
    raw = [synth.make_batch(args.n, mode=args.mode, n_reads=args.n_reads, sub_rate=args.sub_rate, first=i * args.n) for i in range(args.distinct)]
    chroms, batches = one_genome(raw)
    ctx = refine.Context(params=params)
    ctx.set_chromosomes(chroms)
    if args.torch >= 3:
        for b in batches[:4]:
            rb = ctx.upload(b)
            rb.run(keep[0].cuda_stream)
            keep.append(rb)
        import torch
        torch.cuda.synchronize()
    out = {}
    depths = [int(x) for x in args.depths.split(",")] if args.depths else ([args.only_depth] if args.only_depth else sorted(set([1, 2, 3, args.depth])))
    for depth in depths:
        out["depth_%d%s" % (depth, "" if ("depth_%d" % depth) not in out else "_again")] = stream_rate(ctx, batches, raw[0].with_msa, depth, args.batches)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
