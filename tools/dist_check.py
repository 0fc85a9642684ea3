"""The RCCL leg of bench.py exercised with world_size 1 (the GPU box has one GPU): device-resident result records -> torch tensor
-> all_gather_into_tensor on the launch stream; the gathered bytes must equal fetch()."""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
import torch, torch.distributed as dist, numpy as np
import bench
from delly_amd import refine, synth, abi
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
batch = synth.make_batch(2000, mode="c2")
ctx = refine.Context(device=0); ctx.set_chromosomes(batch.chroms)
rb = ctx.upload(batch)
side = torch.cuda.Stream(device=0)
ptr, nbytes = rb.device_results()
res_t = torch.as_tensor(bench._DevPtr(ptr, nbytes), device="cuda:0")
gathered = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    with torch.cuda.stream(side):
        rb.run(side.cuda_stream)
        dist.all_gather_into_tensor(gathered, res_t)
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
res, _ = rb.fetch()
got = np.frombuffer(gathered.cpu().numpy().tobytes(), dtype=abi.result_dtype())
same = all((got[f] == res[f]).all() for f in res.dtype.names if not f.endswith("_off"))   # (fetch() compacts the blob: offsets differ by design)
print("RCCL gather of device-resident records identical to fetch() (all fields but the blob offsets):", same, "ok", int(res["ok"].sum()))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("all_reduce", float(t.item()))
dist.destroy_process_group()
