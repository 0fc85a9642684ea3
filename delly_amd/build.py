"""Builds delly_amd/libdellyhip.so (hand-written HIP for gfx950 + the C-ABI host
side) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdellyhip.so")
SOURCES = ["dellyhip.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))) + ["../../include/dellyhip.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False, out=None, extra_flags=()):
    """out / extra_flags: tuning experiments only (e.g. -DDH_QUAD_WAVES=3 into a side file)"""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-Wno-unused-value",  # profile-Gotoh scores must round like the reference
           "-o", out or LIB] + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]   # (dlopen of RCCL at the first multi-GPU gather)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out or LIB


if __name__ == "__main__":
    build_lib(force=True, verbose=True)


# The sources the headline's kernel (split_sparse_kernel) is compiled from.  profiles/rNN/pmc_traffic.json is stamped with their
# hash when the counters are collected; bench.py compares it with the tree it runs from (`roofline.traffic_stale`).
HEADLINE_KERNEL_SOURCES = ("sparse_needle.hpp", "split_sparse.hpp", "split_main.hpp", "split_kernel.hpp")


def headline_kernel_hash():
    """hash of the CODE of the headline kernel's sources: `//` comments, trailing blanks and empty lines do not count (a stamp
    that a reworded comment invalidates costs a GPU pass to renew)"""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in HEADLINE_KERNEL_SOURCES:
        with open(os.path.join(CSRC, f), "r", encoding="utf-8", errors="replace") as fh:
            for ln in fh:
                ln = re.sub(r"//.*$", "", ln).rstrip()
                if ln:
                    h.update(ln.encode("utf-8") + b"\n")
    return h.hexdigest()[:16]
