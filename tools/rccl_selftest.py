#!/usr/bin/env python
"""RCCL self-test for a box with >= 2 GPUs: no box the builder could reach had one, so RcclLink with more than one
rank -- ncclAllGather of the sizes, grouped ncclSend / ncclRecv of records and consensus bytes over xGMI (delly_amd/csrc/comm.hpp) --
has only ever carried a one-rank communicator.  This runs it for real in under a minute:

    python tools/rccl_selftest.py            # 2 ranks, one per GPU, 128 junctions per rank and step

bench.py --gpus 2 starts the ranks (torch.distributed.run, rendezvous on 127.0.0.1), every step returns the previous step's results
to rank 0 twice over (the blocking RCCL gather and the per-rank shared-memory return), rank 0 dumps what it holds after the last
step of each path, and this script compares both with the reference's own code (oracle/_ref when built, else the C restatement)
on the same junctions -- records, consensus bytes, and the REF,ALT alleles re-cut from the compact payload.  Exit code 0 = identical.
It is the body of tests/test_gpu_multirank.py::test_two_ranks_two_devices_rccl (skipped on one-GPU boxes) as a stand-alone tool."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import numpy as np
    import torch
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n, steps, warm = 128, 3, 1
    if torch.cuda.device_count() < world:
        print("rccl_selftest: %d GPU(s) visible, %d needed -- nothing run (RCCL refuses ranks that share a device; the one-device "
              "protocol test is tests/test_gpu_multirank.py)" % (torch.cuda.device_count(), world))
        return 2
    import bench
    import pyoracle
    from delly_amd import synth
    from util import compare_compact
    pyoracle.build()
    checker = pyoracle.Oracle("reference" if pyoracle.have_reference() else "port")
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "rank0.npz")
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warm),
                            "--junctions", str(n), "--repeats", "1", "--no-cpu-baseline", "--no-extras", "--no-host-inclusive", "--dump-rank0-view", out],
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if p.returncode != 0:
            print(p.stderr[-4000:])
            print("rccl_selftest: bench.py --gpus %d failed (exit code %d)" % (world, p.returncode))
            return 1
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        view = np.load(out)
        cfg = line["config"]
        assert line["n_gpus"] == world and cfg["ranks_that_ran_kernels"] == world, cfg
        assert cfg["gather_transport"] == "rccl" and cfg["rccl_ranks"] == world and not cfg["oversubscribed_one_device"], cfg
        assert cfg["gathered_records_on_rank0"] == world * n and cfg["shm_return_records_seen_by_rank0"] == world * n, cfg
        idx = (warm + steps - 2) % bench.MULTI_RESIDENT_BATCHES       # step k returns the batch of step k - 1
        rec, blob = view["rccl_records"], view["rccl_blob"]
        for r in range(world):
            raw = [synth.make_batch(n, mode="c2", first=(k * world + r) * n) for k in range(bench.MULTI_RESIDENT_BATCHES)]
            _, batches = bench.one_genome(synth, raw)
            b = batches[idx]
            rr, rb = checker.refine_batch(b, want_alignment=False, n_threads=os.cpu_count() or 1)
            compare_compact(rec[r * n:(r + 1) * n], blob, rr, rb, b, label="RCCL-gathered share of rank %d" % r)
            compare_compact(view["shm_records_%d" % r], view["shm_blob_%d" % r], rr, rb, b, label="shared-memory segment of rank %d" % r)
    print("rccl_selftest ok: %d ranks x %d junctions, RCCL gather and shared-memory return identical to the %s checker; gather %.3f ms per step, "
          "value %.2f M alignments/s (%.1f s)" % (world, n, checker.kind, cfg["gather_ms_per_step"], line["value"] / 1e6, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
