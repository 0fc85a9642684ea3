"""Junction sharding across the GPUs of one node.

Junctions are independent (each CPU task of src/shortpe.h:183-197 touches only
its own svs[svid]), so the path shards by junction index with NO data-path
collective; the only exchange is the gather of the small per-junction result
records to the rank that emits the VCF.  With torch.distributed the backend
"nccl" is RCCL on ROCm (xGMI); the same code runs on "gloo" for CPU tests.
"""
import numpy as np

from . import abi


def shard_range(n_total, rank, world):
    """Contiguous block partition of junction indices 0..n_total-1 -> (first, count)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


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
