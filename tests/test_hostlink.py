"""The collective steps of dellyhip_gather_results on the shared-memory transport, WITHOUT a GPU: two (three) real
processes load libdellyhip.so, meet in a hostlink communicator (dellyhip_comm_create_hostlink, ctx = NULL) and run the
size exchange and the root-ready exchange -- the abort protocol of CHANGELOG.md 5: a rank that reports a failure (count = ~0
on the wire) must make EVERY rank leave with an error before anybody posts a payload, and nobody may hang."""
import multiprocessing as mp
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(name, rank, world, script, q, start_delay=0.0):
    try:
        os.environ["DELLYHIP_LINK_TIMEOUT_S"] = "20"
        time.sleep(start_delay)
        from delly_amd import refine
        comm = refine.Comm(None, rank, world, hostlink=name)
        out = []
        for step in script:
            kind = step[0]
            try:
                if kind == "sizes":
                    fail_rank = step[1]
                    out.append(("ok", comm.exchange_sizes(100 + rank, 1000 * (rank + 1), failed=(rank == fail_rank))))
                elif kind == "ready":
                    comm.exchange_ready(root=step[1], root_failed=step[2])
                    out.append(("ok", None))
                elif kind == "gather":
                    payload = bytes([65 + rank]) * step[1][rank]
                    out.append(("ok", comm.gather_bytes(payload, root=step[2], cap=step[3])))
                elif kind == "info":
                    out.append(("ok", comm.info()))
                elif kind == "sleep":
                    time.sleep(step[1].get(rank, 0.0))
                    out.append(("ok", None))
            except refine.DellyHipError as e:
                out.append(("err", e.code, str(e)))
        comm.close()
        q.put((rank, out))
    except Exception as e:   # pragma: no cover
        q.put((rank, [("crash", repr(e))]))


def _run(world, script, timeout=60, name=None, start_delay=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = name or "t%d_%d" % (os.getpid(), int(time.time() * 1e3) % 100000000)
    ps = [ctx.Process(target=_worker, args=(name, r, world, script, q, (start_delay or {}).get(r, 0.0))) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    t0 = time.time()
    while len(got) < world and time.time() - t0 < timeout:
        try:
            r, out = q.get(timeout=1.0)
            got[r] = out
        except Exception:
            pass
    for p in ps:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
    assert len(got) == world, "a rank hung or died: %r" % (got,)
    return got


def test_size_exchange_all_ranks_see_all_pairs():
    got = _run(3, [("info",), ("sizes", -1), ("sizes", -1), ("ready", 0, False)])
    for r in range(3):
        info, s1, s2, rd = got[r]
        assert info[0] == "ok" and info[1]["kind"] == "hostlink" and info[1]["world"] == 3 and info[1]["rank"] == r
        assert s1 == ("ok", [(100, 1000), (101, 2000), (102, 3000)]) and s2 == s1   # (two rounds: both parity buffers)
        assert rd == ("ok", None)


def test_abort_protocol_a_failing_rank_makes_every_rank_fail_and_nobody_hangs():
    from delly_amd import abi
    got = _run(2, [("sizes", 1), ("sizes", -1)])
    # round 1: rank 1 reports a failure -> both ranks return an error; the failing rank keeps its own message, the other names it
    assert got[0][0][0] == "err" and got[0][0][1] == abi.E_RUNTIME and "rank 1 failed before the exchange" in got[0][0][2]
    assert got[1][0][0] == "err" and got[1][0][1] == abi.E_RUNTIME and "this rank reported a failure" in got[1][0][2]
    # round 2: the communicator is still usable (the protocol stayed in step)
    assert got[0][1] == ("ok", [(100, 1000), (101, 2000)]) and got[1][1] == got[0][1]


def test_abort_protocol_root_not_ready():
    from delly_amd import abi
    got = _run(2, [("sizes", -1), ("ready", 0, True), ("ready", 1, True), ("ready", 0, False)])
    for r in range(2):
        assert got[r][0][0] == "ok"
        assert got[r][1][0] == "err" and got[r][1][1] == abi.E_NOMEM and "root could not allocate" in got[r][1][2]
        assert got[r][2][0] == "err" and got[r][2][1] == abi.E_NOMEM      # (any root)
        assert got[r][3] == ("ok", None)


def test_payload_group_gatherv_of_ragged_sizes_growing_outbox():
    # ragged payloads incl. an empty one, a second round larger than the first outbox (new generation), another root
    sizes1, sizes2 = [5, 0, 70000], [3 << 20, 11, 1]
    got = _run(3, [("gather", sizes1, 0, 1 << 24), ("gather", sizes2, 0, 1 << 24), ("gather", sizes1, 2, 1 << 24), ("gather", sizes2, 0, 16)])
    for sizes, root, k in ((sizes1, 0, 0), (sizes2, 0, 1), (sizes1, 2, 2)):
        want = b"".join(bytes([65 + r]) * sizes[r] for r in range(3))
        for r in range(3):
            tag, (data, sz) = got[r][k]
            assert tag == "ok" and sz == sizes
            assert data == (want if r == root else None)
    from delly_amd import abi
    for r in range(3):   # the root's buffer is too small: everybody learns it before any payload moves
        assert got[r][3][0] == "err" and got[r][3][1] == abi.E_NOMEM


def test_alternating_roots_with_a_slow_sender():
    # the message sequence must be per (sender, destination).  Root 0 twice (rank 1 and 2 each publish two
    # messages for rank 0), then root 2 while rank 1 is slow: rank 2 reaches its receive for rank 1 long before rank 1
    # publishes, and must not mistake rank 1's earlier messages (for rank 0) for its own.  Then root 1, then root 0 again.
    sizes = [[7, 300, 5000], [1 << 20, 9, 2], [11, 70000, 13], [5, 6, 7], [100, 200, 300]]
    roots = [0, 0, 2, 1, 0]
    script = []
    for k, (sz, root) in enumerate(zip(sizes, roots)):
        if k == 2:
            script.append(("sleep", {0: 0.3, 1: 1.5}))
        if k == 3:
            script.append(("sleep", {2: 1.0}))
        script.append(("gather", sz, root, 1 << 24))
    got = _run(3, script)
    for r in range(3):
        outs = [o for o in got[r] if not (o[0] == "ok" and o[1] is None)]
        assert len(outs) == len(sizes), got[r]
        for k, (sz, root) in enumerate(zip(sizes, roots)):
            want = b"".join(bytes([65 + q]) * sz[q] for q in range(3))
            tag, (data, got_sz) = outs[k]
            assert tag == "ok" and got_sz == sz
            assert data == (want if r == root else None)


def test_leftover_segment_of_a_run_that_crashed_before_the_peer_attached():
    # a control segment left by a run that died after rank 0 attached and before rank 1 did looks fresh to rank 1
    # (MAGIC set, its own slot untouched, rank 0 not marked "left").  Rank 1 starts first and attaches to the corpse; rank 0 of the
    # new run then unlinks it and creates the live segment: rank 1 must move over instead of waiting out its deadline.
    import struct
    name = "stale%d_%d" % (os.getpid(), int(time.time() * 1e3) % 100000000)
    path = "/dev/shm/dellyhip_%s_ctl" % name
    slot = 8 + 8 + 32 + 64 * 8 + 64 * 8 + 8 + 8 + 8 * 8 + 64     # HostLink::Slot (comm.hpp)
    corpse = bytearray(16 + 2 * slot)
    struct.pack_into("<QQQ", corpse, 0, 0x64656c6c79686c31, 2, 1)  # magic, world, slot[0].attached = 1
    with open(path, "wb") as f:
        f.write(corpse)
    try:
        t0 = time.time()
        got = _run(2, [("sizes", -1), ("gather", [10, 2000], 0, 1 << 20)], name=name, start_delay={0: 1.5})
        assert time.time() - t0 < 15, "rank 1 waited for its deadline instead of following the replaced segment"
        for r in range(2):
            assert got[r][0] == ("ok", [(100, 1000), (101, 2000)]), got[r]
            assert got[r][1][0] == "ok"
        assert got[0][1][1][0] == b"A" * 10 + b"B" * 2000
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_a_missing_peer_is_an_error_not_a_hang():
    # rank 1 never shows up: rank 0's first exchange must time out with an error
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["DELLYHIP_LINK_TIMEOUT_S"] = "1"
    try:
        name = "lonely%d" % os.getpid()
        p = ctx.Process(target=_worker_short, args=(name, q))
        p.start()
        rank, out = q.get(timeout=60)
        p.join(timeout=10)
    finally:
        os.environ.pop("DELLYHIP_LINK_TIMEOUT_S", None)
    assert out[0][0] == "err" and "timed out waiting for rank 1" in out[0][2]


def _worker_short(name, q):
    os.environ["DELLYHIP_LINK_TIMEOUT_S"] = "1"
    from delly_amd import refine
    comm = refine.Comm(None, 0, 2, hostlink=name)
    try:
        comm.exchange_sizes(1, 1)
        q.put((0, [("ok",)]))
    except refine.DellyHipError as e:
        q.put((0, [("err", e.code, str(e))]))
    comm.close()
