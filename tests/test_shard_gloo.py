"""CPU, world_size 2, gloo: the N>1 path of bench.py / tests/shard_mirror.py --
junction sharding by index + all-gather of the fixed-size result records --
reassembles exactly the single-process result.  (The per-rank compute is the
C oracle here: the product has no CPU path.)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard.shard_range(n_total, rank, world)
    b = synth.make_batch(count, mode="mixed", seed=9, first=first)
    res, _ = pyoracle.Oracle("port").refine_batch(b, want_alignment=False)
    local = torch.from_numpy(np.frombuffer(res.tobytes(), dtype=np.uint8).copy())
    gathered, counts, mx = shard.gather_records(local, world, dist)
    if rank == 0:
        merged = shard.merge_records(gathered.numpy(), counts, mx)
        np.save(os.path.join(out_dir, "merged.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import synth
    n_total = 37  # odd: ranks hold different counts -> padded gather
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    merged = np.load(os.path.join(str(tmp_path), "merged.npy"))
    whole = synth.make_batch(n_total, mode="mixed", seed=9, first=0)
    ref, _ = pyoracle.Oracle("port").refine_batch(whole, want_alignment=False)
    assert merged.shape == ref.shape
    for f in ["svid", "ok", "ci_wiggle", "hom_len", "cons_bp", "sr_align_quality", "ins_len"]:
        assert np.array_equal(merged[f], ref[f]), f
    # a shard's private chromosome starts at its first junction: coordinates are
    # relative to that origin (synth.WINDOW bases per junction)
    first = np.array([shard.shard_range(n_total, r, 2)[0] for r in (0, 1) for _ in range(shard.shard_range(n_total, r, 2)[1])])
    assert np.array_equal(merged["sv_start"] + first * synth.WINDOW, ref["sv_start"])
    assert np.array_equal(merged["sv_end"] + first * synth.WINDOW, ref["sv_end"])


def _job_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import abi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs, blob = synth.make_align_jobs(9, 7, seed=4)          # every rank sees the job list, classifies its block
    first, count = shard.shard_range(jobs.shape[0], rank, world)
    res = pyoracle.Oracle("port").classify_reads(jobs[first:first + count], blob)
    local = torch.from_numpy(np.frombuffer(res.tobytes(), dtype=np.uint8).copy())
    rec = abi.align_result_dtype().itemsize
    gathered, counts, mx = shard.gather_records(local, world, dist, record_bytes=rec)
    if rank == 0:
        merged = shard.merge_records(gathered.numpy(), counts, mx, dtype=abi.align_result_dtype(), sort_key=None)
        np.save(os.path.join(out_dir, "jobs.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_classifier_jobs_equal_single_process(tmp_path):
    """the genotyping rows shard by job index exactly like junctions: block partition + padded all-gather"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    from delly_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_job_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    merged = np.load(os.path.join(str(tmp_path), "jobs.npy"))
    jobs, blob = synth.make_align_jobs(9, 7, seed=4)
    whole = pyoracle.Oracle("port").classify_reads(jobs, blob)
    assert merged.tobytes() == whole.tobytes()


def _cost_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole = synth.make_batch(30, mode="mixed", n_reads=5, seed=13)          # every rank sees the junction list
    mine = shard.shard_by_cost(whole, rank, world)                          # ... and takes its cost-balanced share
    sub = synth.subset(whole, mine)
    res, blob = pyoracle.Oracle("port").refine_batch(sub, want_alignment=False)
    recs, gblob, counts = shard.gather_results(res, blob, world, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "recs.npy"), recs)
        np.save(os.path.join(out_dir, "blob.npy"), gblob)
        np.save(os.path.join(out_dir, "counts.npy"), np.asarray(counts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cost_sharding_and_gather_of_records_and_blob_bytes(tmp_path):
    """cost-balanced assignment (dellyhip_shard_by_cost) + gather of the records AND the consensus / allele bytes to
    rank 0 (the torch mirror of dellyhip_gather_results): sorted by svid it is the single-process result, byte for byte"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_cost_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs = np.load(os.path.join(str(tmp_path), "recs.npy"))
    blob = np.load(os.path.join(str(tmp_path), "blob.npy"))
    counts = np.load(os.path.join(str(tmp_path), "counts.npy"))
    whole = synth.make_batch(30, mode="mixed", n_reads=5, seed=13)
    ref, rblob = pyoracle.Oracle("port").refine_batch(whole, want_alignment=False)
    assert int(counts.sum()) == whole.n and min(counts) > 0
    order = np.argsort(recs["svid"], kind="stable")
    recs = recs[order]
    assert np.array_equal(recs["svid"], ref["svid"])
    for f in ("ok", "sv_start", "sv_end", "ci_wiggle", "ins_len", "cons_bp", "hom_len", "sr_support", "sr_align_quality",
              "cons_len", "allele_len"):
        assert np.array_equal(recs[f], ref[f]), f
    n_bytes = 0
    for k in range(whole.n):
        for w in ("cons", "allele"):
            a, b = pyoracle.blob_field(recs[k], blob, w), pyoracle.blob_field(ref[k], rblob, w)
            assert a == b, (k, w)
            n_bytes += len(a)
    assert n_bytes > 1000


def test_shard_by_cost_balances_and_is_deterministic():
    from delly_amd import abi, refine, synth
    # 10x cost spread: a few long-read junctions among short ones
    b = synth.make_batch(40, mode="c2", n_reads=8, seed=3)
    cost = (b.junctions["n_seq"].astype(np.float64)) ** 2
    for world in (1, 2, 3, 8):
        o1 = refine.shard_by_cost(b.junctions, b.seq_off, world)
        o2 = refine.shard_by_cost(b.junctions, b.seq_off, world)
        assert np.array_equal(o1, o2) and o1.min() >= 0 and o1.max() < world
        loads = np.array([cost[o1 == r].sum() for r in range(world)])
        assert loads.max() <= loads.mean() * 1.25 + cost.max()
    lr = synth.make_batch(6, mode="lr", n_reads=5, sub_rate=0.05)
    mixed_j = np.concatenate([b.junctions[:20], lr.junctions])
    mixed_j["seq_first"][20:] += b.n_seq
    off = np.concatenate([b.seq_off, lr.seq_off[1:] + b.seq_off[-1]])
    own = refine.shard_by_cost(mixed_j, off, 2, abi.params_lr())
    # the six expensive junctions are split 3 / 3, not left on one rank as a contiguous block would
    assert sorted(np.bincount(own[20:], minlength=2).tolist()) == [3, 3]


def test_shard_range_partitions():
    import shard_mirror as shard
    for n in (0, 1, 7, 64, 10001):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                f, c = shard.shard_range(n, r, w)
                seen.extend(range(f, f + c))
            assert seen == list(range(n))


def _shm_worker(rank, world, port, n_total, out_dir):
    """every rank writes its own results into its shared-memory segment (what dellyhip_batch_fetch does on the GPU box:
    bench.py --gpus N), rank 0 maps all of them and merges in place -- no collective carries results"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import abi, shmreturn, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rb = abi.result_dtype().itemsize
    seg = shmreturn.Segment("t%d" % port, rank, 64, rb, 1 << 20, create=True)
    dist.barrier()
    others = [shmreturn.Segment("t%d" % port, r, 64, rb, 1 << 20, create=False) for r in range(world) if r != rank] if rank == 0 else []
    first, count = shard.shard_range(n_total, rank, world)
    for lap in range(3):   # three batches through the same segment; the last one is what rank 0 merges
        b = synth.make_batch(count, mode="mixed", seed=9 + 2 - lap, first=first)
        res, blob = pyoracle.Oracle("port").refine_batch(b, want_alignment=False)
        seg.begin()
        assert seg.read(abi.result_dtype()) is None            # a reader never sees a half-written batch
        seg.records_view()[:res.nbytes] = np.frombuffer(res.tobytes(), dtype=np.uint8)
        seg.blob_view()[:blob.nbytes] = blob
        seg.commit(count, blob.nbytes)
    dist.barrier()
    if rank == 0:
        parts = []
        for sg in [seg] + others:
            seqno, rec, bl = sg.read(abi.result_dtype())
            assert seqno == 3
            parts.append((rec.copy(), bl.copy()))
        np.save(os.path.join(out_dir, "shm_records.npy"), np.concatenate([p[0].view(np.uint8) for p in parts]))
        np.save(os.path.join(out_dir, "shm_blob_sizes.npy"), np.array([p[1].nbytes for p in parts]))
        np.save(os.path.join(out_dir, "shm_blob.npy"), np.concatenate([p[1] for p in parts]))
    for sg in others:
        sg.close()
    dist.barrier()
    seg.close()
    dist.destroy_process_group()


def test_two_rank_return_through_shared_memory_segments(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import shard_mirror as shard
    from delly_amd import abi, synth
    n_total = 29
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shm_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    rec = np.load(os.path.join(str(tmp_path), "shm_records.npy")).view(abi.result_dtype())
    sizes = np.load(os.path.join(str(tmp_path), "shm_blob_sizes.npy"))
    blob = np.load(os.path.join(str(tmp_path), "shm_blob.npy"))
    at, base = 0, 0
    for r in (0, 1):
        first, count = shard.shard_range(n_total, r, 2)
        b = synth.make_batch(count, mode="mixed", seed=9, first=first)
        ref, rblob = pyoracle.Oracle("port").refine_batch(b, want_alignment=False)
        mine = rec[at:at + count]
        for f in ref.dtype.names:
            assert np.array_equal(mine[f], ref[f]), (r, f)
        assert blob[base:base + sizes[r]].tobytes() == rblob.tobytes()
        at += count
        base += int(sizes[r])
    assert at == rec.shape[0]
