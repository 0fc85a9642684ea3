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
