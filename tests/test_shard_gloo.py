"""CPU, world_size 2, gloo: the N>1 path of bench.py / delly_amd.shard --
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
    import pyoracle
    from delly_amd import shard, synth
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
    import pyoracle
    from delly_amd import shard, synth
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
    import pyoracle
    from delly_amd import abi, shard, synth
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


def test_shard_range_partitions():
    from delly_amd import shard
    for n in (0, 1, 7, 64, 10001):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                f, c = shard.shard_range(n, r, w)
                seen.extend(range(f, f + c))
            assert seen == list(range(n))
