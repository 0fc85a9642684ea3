"""CPU: the C-ABI library builds, loads and exports every symbol that
include/dellyhip.h declares; struct layouts of the ctypes mirror match.  No
compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from delly_amd import abi, build, refine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return refine.load_library()


def _declared():
    txt = open(os.path.join(ROOT, "include", "dellyhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dellyhip_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libdellyhip.so does not export %s" % n
    assert set(refine.EXPORTS) == set(names)


def test_struct_layouts(lib):
    out = (C.c_int32 * 4)()
    lib.dellyhip_abi_info.restype = None
    lib.dellyhip_abi_info(out)
    assert out[0] == 1
    assert out[1] == C.sizeof(abi.Params)
    assert out[2] == C.sizeof(abi.Junction)
    assert out[3] == C.sizeof(abi.Result)
    assert abi.result_dtype().itemsize == C.sizeof(abi.Result)
    assert abi.junction_dtype().itemsize == C.sizeof(abi.Junction)
    # dellyhip_align_job / dellyhip_align_result (static_assert'ed to these sizes in csrc/dellyhip.hip)
    assert C.sizeof(abi.AlignJob) == 48 and abi.align_job_dtype().itemsize == 48
    assert C.sizeof(abi.AlignResult) == 20 and abi.align_result_dtype().itemsize == 20


def test_no_cpu_fallback(lib):
    """Without a GPU the product must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(refine.DellyHipError) as e:
        refine.Context()
    assert e.value.code == abi.E_NODEVICE


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under delly_amd/ may reference it."""
    pkg = os.path.join(ROOT, "delly_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "delly_oracle" not in txt, f
                assert "libdelly_ref" not in txt, f


def test_compiled_cpp_caller_builds_and_fails_loudly_without_a_device():
    """tests/cpp/dropin_test: the drop-in headers compile against the reference's tags.h / align.h, the struct layouts
    agree with the library, and without a usable GPU torali::msa() throws (no CPU path)."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "dropin_test")
    if not os.path.exists(exe):
        if not os.path.isdir("/root/reference/src"):
            pytest.skip("tests/cpp/_build/dropin_test not built and /root/reference absent")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe, "abi"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "nodevice"], capture_output=True, text=True)
        assert r.returncode == 0 and "loud failure" in r.stdout, r.stdout + r.stderr


def test_rebase_of_gathered_records_multi_rank_on_the_host():
    """dellyhip_rebase_gathered = what the root of dellyhip_gather_results does to the records of W ranks after the RCCL
    exchange (offset rebasing across the concatenated per-rank blobs).  Pure host code: driven here with faked multi-rank
    inputs, no GPU and no communicator needed."""
    import ctypes as C

    import numpy as np

    from delly_amd import abi, refine
    lib = refine.load_library()
    rng = np.random.default_rng(5)
    world = 5
    recs, blobs, counts, nbytes, want = [], [], [], [], []
    for r in range(world):
        n = int(rng.integers(0, 40)) if r != 2 else 0          # one rank without junctions
        rec = np.zeros(n, dtype=abi.result_dtype())
        parts = []
        at = 0
        for k in range(n):
            lens = [int(rng.integers(0, 300)), int(rng.integers(0, 900)) if rng.random() < 0.7 else 0, int(rng.integers(0, 200)) if rng.random() < 0.3 else 0]
            rec[k]["cons_len"], rec[k]["allele_len"], rec[k]["aln_len"] = lens
            rec[k]["svid"] = 1000 * r + k
            rec[k]["reserved"] = 2                              # transient kernel state must not survive
            pieces = [rng.integers(65, 90, lens[0], dtype=np.uint8), rng.integers(65, 90, lens[1], dtype=np.uint8),
                      rng.integers(65, 90, 2 * lens[2], dtype=np.uint8)]
            # what a rank's compaction leaves: offsets relative to ITS blob (any value where the length is 0)
            rec[k]["cons_off"], rec[k]["allele_off"], rec[k]["aln_off"] = at, at + lens[0], at + lens[0] + lens[1]
            at += lens[0] + lens[1] + 2 * lens[2]
            parts.extend(pieces)
            want.append([p.tobytes() for p in pieces])
        recs.append(rec)
        blobs.append(np.concatenate(parts) if parts else np.zeros(0, np.uint8))
        counts.append(n)
        nbytes.append(at)
    allr = np.zeros(sum(counts), dtype=abi.result_dtype())   # (np.concatenate would re-pack the padded record layout)
    at = 0
    for rec in recs:
        allr[at:at + rec.shape[0]] = rec
        at += rec.shape[0]
    blob = np.concatenate(blobs)
    cnt = np.array(counts, dtype=np.uint64)
    byt = np.array(nbytes, dtype=np.uint64)
    rc = lib.dellyhip_rebase_gathered(allr.ctypes.data_as(C.c_void_p), C.c_uint64(allr.shape[0]), world,
                                      cnt.ctypes.data_as(C.POINTER(C.c_uint64)), byt.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0, lib.dellyhip_last_error()
    assert (allr["reserved"] == 0).all()
    for k in range(allr.shape[0]):
        R = allr[k]
        for off, ln, w in ((int(R["cons_off"]), int(R["cons_len"]), want[k][0]), (int(R["allele_off"]), int(R["allele_len"]), want[k][1]),
                           (int(R["aln_off"]), 2 * int(R["aln_len"]), want[k][2])):
            assert blob[off:off + ln].tobytes() == w, (k, off, ln)
    # inconsistent inputs are refused: a blob size that does not match the records, counts that do not cover them
    byt2 = byt.copy()
    byt2[1] += 1
    assert lib.dellyhip_rebase_gathered(allr.ctypes.data_as(C.c_void_p), C.c_uint64(allr.shape[0]), world,
                                        cnt.ctypes.data_as(C.POINTER(C.c_uint64)), byt2.ctypes.data_as(C.POINTER(C.c_uint64))) != 0
    assert lib.dellyhip_rebase_gathered(allr.ctypes.data_as(C.c_void_p), C.c_uint64(allr.shape[0] + 1), world,
                                        cnt.ctypes.data_as(C.POINTER(C.c_uint64)), byt.ctypes.data_as(C.POINTER(C.c_uint64))) != 0
