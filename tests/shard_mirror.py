"""TEST-ONLY torch.distributed mirror of the junction sharding / result gather of one node.  The product's
gather lives in the host library (dellyhip_gather_results, delly_amd/csrc/comm.hpp); this mirror lets the protocol's
arithmetic -- block partition, padded all-gather, offset rebasing -- run on gloo without a GPU (tests/test_shard_gloo.py).

Junctions are independent (each CPU task of src/shortpe.h:183-197 touches only
its own svs[svid]), so the path shards by junction index with NO data-path
collective; the only exchange is the gather of the small per-junction result
records to the rank that emits the VCF.  With torch.distributed the backend
"nccl" is RCCL on ROCm (xGMI); the same code runs on "gloo" for CPU tests.
"""
import numpy as np

from delly_amd import abi


def shard_range(n_total, rank, world):
    """Contiguous block partition of junction indices 0..n_total-1 -> (first, count)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def shard_by_cost(batch, rank, world, params=None):
    """Cost-balanced junction assignment (dellyhip_shard_by_cost in the host library): indices of the
    junctions rank `rank` refines.  Every rank computes the same assignment from the same junction list."""
    from delly_amd import refine
    owner = refine.shard_by_cost(batch.junctions, batch.seq_off, world, params)
    return np.nonzero(owner == rank)[0]


def gather_results(local_rec, local_blob, world, dist=None):
    """torch.distributed mirror of dellyhip_gather_results (the product path uses RCCL directly inside the host library;
    this one runs on gloo for the CPU tests): all-gathers the fixed-size records and the variable-length consensus /
    allele bytes, rebases the blob offsets.  local_rec: structured array (abi.result_dtype) whose *_off fields point into
    local_blob (np.uint8, compact).  -> (records in rank order, blob, counts)"""
    import torch
    dt = abi.result_dtype()
    rec_t = torch.from_numpy(np.frombuffer(np.ascontiguousarray(local_rec).tobytes(), dtype=np.uint8).copy())
    if world == 1:
        return local_rec.copy(), np.asarray(local_blob, dtype=np.uint8).copy(), [int(local_rec.shape[0])]
    gathered, counts, mx = gather_records(rec_t, world, dist)
    recs = merge_records(gathered.cpu().numpy(), counts, mx, sort_key=None).copy()
    blob_t = torch.from_numpy(np.ascontiguousarray(local_blob, dtype=np.uint8).copy())
    gb, bcounts, bmx = gather_records(blob_t, world, dist, record_bytes=1)
    gb = gb.cpu().numpy().reshape(world, bmx)
    blob = np.concatenate([gb[r, :bcounts[r]] for r in range(world)]) if world else np.zeros(0, np.uint8)
    base = np.concatenate([[0], np.cumsum(bcounts)])[:-1]
    pos = 0
    for r, c in enumerate(counts):
        seg = recs[pos:pos + c]
        for f, ln, mult in (("cons_off", "cons_len", 1), ("allele_off", "allele_len", 1), ("aln_off", "aln_len", 2)):
            has = seg[ln] > 0
            seg[f][has] += np.uint64(base[r])
        pos += c
    return recs, blob, counts


def gather_records(local, world, dist=None, max_count=None, record_bytes=None):
    """All-gathers fixed-size result records (a uint8 torch tensor of
    count*sizeof(dellyhip_result) bytes, on the device of the backend).  Ranks
    may hold different counts: records are padded to max_count (RCCL has no
    all-gatherv; payload is ~150 B per junction, latency-bound).  record_bytes: size of one record when it is not
    a dellyhip_result (the genotyping rows shard the same way: 20-byte dellyhip_align_result, 4-byte distances)."""
    import torch
    rec = record_bytes or abi.result_dtype().itemsize
    count = local.numel() // rec
    if world == 1:
        return local, [count], count
    if max_count is None:
        t = torch.tensor([count], dtype=torch.int64, device=local.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_count = int(t.item())
    pad = torch.zeros(max_count * rec, dtype=torch.uint8, device=local.device)
    pad[:local.numel()] = local
    out = torch.empty(world * max_count * rec, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    counts = torch.tensor([count], dtype=torch.int64, device=local.device)
    allc = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(allc, counts)
    return out, [int(x) for x in allc.cpu()], max_count


def merge_records(gathered_bytes, counts, max_count, dtype=None, sort_key="svid"):
    """Host side: strips the padding and orders by svid (the CPU result is
    order-independent because every task writes its own slot).  sort_key=None keeps rank order
    (block-partitioned job lists: rank order IS job order)."""
    dt = dtype or abi.result_dtype()
    arr = np.frombuffer(np.ascontiguousarray(gathered_bytes), dtype=dt).reshape(len(counts), max_count)
    parts = [arr[r, :c] for r, c in enumerate(counts)]
    allr = np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
    if sort_key is None:
        return allr
    return allr[np.argsort(allr[sort_key], kind="stable")]
