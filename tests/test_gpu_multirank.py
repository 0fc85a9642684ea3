"""-m gpu: the N > 1 path with TWO REAL PROCESSES on the one GPU of the test box.

bench.py --gpus 2 --oversubscribe starts its two ranks itself (torch.distributed.run), both drive device 0, and every step
returns the previous step's results to rank 0 twice over: through dellyhip_gather_results -- the same protocol code that runs
on RCCL with one process per GPU, here on the shared-memory transport because RCCL refuses ranks that share a device -- and
through the per-rank shared-memory segments.  What rank 0 holds after the last step of each path is compared, bit for bit,
with the reference's own code (oracle/_ref) on the same junctions.  Also: the launcher contract (a WORLD_SIZE that does not
match --gpus is refused; --gpus beyond the node's devices is refused) and the abort protocol inside a real gather."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from delly_amd import abi, synth
from util import CORE, compare, compare_compact

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N, STEPS, WARM, WORLD = 1500, 3, 1, 2
THREADS = os.cpu_count() or 1
LAST_RETURNED = (WARM + STEPS - 2) % bench.MULTI_RESIDENT_BATCHES   # (steps count from 0; step k returns the batch of step k - 1)


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(kw)
    return env


def _bench(args, env=None, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env or _env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, text=True)
    return p


def _rank_batches(rank, idx):
    """the resident batch `idx` of `rank` exactly as bench.py builds it (weak scaling, one concatenated genome per rank)"""
    raw = [synth.make_batch(N, mode="c2", first=(k * WORLD + rank) * N) for k in range(bench.MULTI_RESIDENT_BATCHES)]
    chroms, batches = bench.one_genome(synth, raw)
    return chroms, batches[idx]


@pytest.fixture(scope="module")
def two_rank_run(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mr") / "rank0.npz")
    p = _bench(["--gpus", str(WORLD), "--oversubscribe", "--steps", str(STEPS), "--warmup", str(WARM), "--junctions", str(N),
                "--no-cpu-baseline", "--no-extras", "--no-host-inclusive", "--dump-rank0-view", out])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]      # the contract: rank 0 prints ONE JSON line
    return json.loads(lines[0]), np.load(out)


def test_two_ranks_started_by_bench_itself_and_counted(two_rank_run):
    line, _ = two_rank_run
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["ranks_launched"] == WORLD and cfg["ranks_that_ran_kernels"] == WORLD   # (n_gpus = distinct devices: both ranks drive device 0)
    assert cfg["rccl_ranks"] == WORLD and cfg["gather_transport"] == "hostlink" and cfg["oversubscribed_one_device"] is True
    assert cfg["gathered_records_on_rank0"] == WORLD * N and cfg["shm_return_records_seen_by_rank0"] == WORLD * N
    assert cfg["gather_ms_per_step"] > 0 and cfg["shm_return_gather_ms_per_step"] > 0
    assert line["value"] > 0 and cfg["shm_return_alignments_per_s"] > 0
    assert abs(line["value"] - WORLD * N * STEPS / (line["ms_per_step"] * 1e-3 * STEPS)) < 1e-6 * line["value"]


def test_what_rank0_holds_after_the_gather_is_the_reference_answer(two_rank_run, reference):
    """dellyhip_gather_results with world = 2: rank order, rebased blob offsets, both ranks' bytes -- vs oracle/_ref"""
    _, view = two_rank_run
    idx = LAST_RETURNED               # the batch refined in the second-to-last step is the one the last step returned
    rec, blob = view["rccl_records"], view["rccl_blob"]
    assert rec.shape[0] == WORLD * N
    for r in range(WORLD):
        chroms, b = _rank_batches(r, idx)
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        mine = rec[r * N:(r + 1) * N]
        # (the N > 1 step returns the compact payload: REF,ALT are re-cut from the record by whoever merges -- here, and compared)
        compare_compact(mine, blob, rr, rb, synth.Batch(chroms, b.junctions, b.seq_blob, b.seq_off, b.with_msa, b.truth), label="gathered share of rank %d" % r)
        assert int(mine["ok"].sum()) >= int(0.98 * N)


def test_what_rank0_reads_from_both_segments_is_the_reference_answer(two_rank_run, reference):
    _, view = two_rank_run
    idx = LAST_RETURNED
    for r in range(WORLD):
        rec, blob = view["shm_records_%d" % r], view["shm_blob_%d" % r]
        assert rec.shape[0] == N
        chroms, b = _rank_batches(r, idx)
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        compare_compact(rec, blob, rr, rb, synth.Batch(chroms, b.junctions, b.seq_blob, b.seq_off, b.with_msa, b.truth), label="segment of rank %d" % r)


def _n_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs: RCCL refuses a communicator whose ranks share a device")
def test_two_ranks_two_devices_rccl(tmp_path, reference):
    """The product transport with more than one rank: bench.py --gpus 2, one process per GPU, NOT oversubscribed -- the gather
    runs RcclLink (ncclAllGather of the sizes, grouped ncclSend / ncclRecv of records and blob bytes over xGMI, comm.hpp).  What
    rank 0 holds after each return path is compared with oracle/_ref like in the one-device test above.  (No box the builder
    could reach in rounds 1-6 had two GPUs: this test is armed for the driver's multi-GPU node; tools/rccl_selftest.py is the same
    check as a stand-alone tool a maintainer can run on any >= 2-GPU box in under a minute.)"""
    out = str(tmp_path / "rank0_rccl.npz")
    p = _bench(["--gpus", str(WORLD), "--steps", str(STEPS), "--warmup", str(WARM), "--junctions", str(N), "--no-cpu-baseline", "--no-extras",
                "--no-host-inclusive", "--dump-rank0-view", out])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line, view = json.loads(lines[0]), np.load(out)
    cfg = line["config"]
    assert line["n_gpus"] == WORLD and cfg["ranks_launched"] == WORLD and cfg["ranks_that_ran_kernels"] == WORLD
    assert cfg["gather_transport"] == "rccl" and cfg["rccl_ranks"] == WORLD and cfg["oversubscribed_one_device"] is False
    assert cfg["gathered_records_on_rank0"] == WORLD * N and cfg["shm_return_records_seen_by_rank0"] == WORLD * N
    idx = LAST_RETURNED
    rec, blob = view["rccl_records"], view["rccl_blob"]
    assert rec.shape[0] == WORLD * N
    for r in range(WORLD):
        chroms, b = _rank_batches(r, idx)
        rr, rb = reference.refine_batch(b, want_alignment=False, n_threads=THREADS)
        bb = synth.Batch(chroms, b.junctions, b.seq_blob, b.seq_off, b.with_msa, b.truth)
        compare_compact(rec[r * N:(r + 1) * N], blob, rr, rb, bb, label="RCCL-gathered share of rank %d" % r)
        compare_compact(view["shm_records_%d" % r], view["shm_blob_%d" % r], rr, rb, bb, label="segment of rank %d (two devices)" % r)


def test_launcher_contract_world_size_must_match_gpus():
    p = _bench(["--gpus", "2", "--steps", "1"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_more_gpus_than_the_node_has_is_refused_not_faked():
    import torch
    p = _bench(["--gpus", str(torch.cuda.device_count() + 1), "--steps", "1"], timeout=120)
    assert p.returncode != 0 and "refused" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


_ABORT = r"""
import os, sys
sys.path.insert(0, %(root)r)
rank, world, name = int(sys.argv[1]), 2, sys.argv[2]
os.environ["DELLYHIP_LINK_TIMEOUT_S"] = "60"
from delly_amd import refine, synth
ctx = refine.Context(device=0)
b = synth.make_batch(64, mode="c2", first=rank * 64)
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
comm = refine.Comm(ctx, rank, world, hostlink=name)
out = []
for inject in ("rank1", "root", None):
    os.environ.pop("DELLYHIP_TEST_FAIL_GATHER_RANK", None); os.environ.pop("DELLYHIP_TEST_FAIL_GATHER_ROOT", None)
    if inject == "rank1": os.environ["DELLYHIP_TEST_FAIL_GATHER_RANK"] = "1"
    if inject == "root": os.environ["DELLYHIP_TEST_FAIL_GATHER_ROOT"] = "1"
    try:
        res, blob, counts = rb.gather(comm, root=0)
        out.append("ok:%%s" %% ("-" if res is None else "%%d:%%s" %% (res.shape[0], ",".join(map(str, counts)))))
    except refine.DellyHipError as e:
        out.append("err:%%d:%%s" %% (e.code, str(e).replace("\n", " ")))
print("RESULT|" + "|".join(out), flush=True)
comm.close(); rb.free(); ctx.close()
"""


def test_abort_protocol_inside_a_real_gather_two_processes_one_gpu(tmp_path):
    """a rank whose batch fails locally, then a root that cannot size its buffers: BOTH ranks must return an error from
    dellyhip_gather_results without hanging, and the next gather on the same communicator must work"""
    script = tmp_path / "abort.py"
    script.write_text(_ABORT % {"root": ROOT})
    name = "abort%d" % os.getpid()
    ps = [subprocess.Popen([sys.executable, str(script), str(r), name], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for r in range(2)]
    outs = [p.communicate(timeout=300) for p in ps]
    got = []
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se[-2000:]
        got.append([ln for ln in so.splitlines() if ln.startswith("RESULT|")][0].split("|")[1:])
    r0, r1 = got
    assert r0[0].startswith("err:%d:" % abi.E_RUNTIME) and "rank 1 failed before the exchange" in r0[0]
    assert r1[0].startswith("err:%d:" % abi.E_RUNTIME) and "injected" in r1[0]
    assert r0[1].startswith("err:%d:" % abi.E_NOMEM) and r1[1].startswith("err:%d:" % abi.E_NOMEM)
    assert r0[2] == "ok:128:64,64" and r1[2] == "ok:-"
