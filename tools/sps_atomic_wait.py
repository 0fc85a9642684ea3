"""Where does split_sparse_kernel wait for its work-counter atomic?  (CHANGELOG.md, round 6: the answer must not be waited for
before split_detect's hook parks it in LDS -- an `s_waitcnt vmcnt(0)` right behind the atomic puts the whole round trip, 1 - 3 us
under load, in front of the stage that is supposed to hide it.)  Compiles the device code of the library with the flags of
delly_amd/build.py and prints, for every atomic of the kernel, how many instructions later the first vmcnt wait comes and what
follows it.  Needs hipcc only:  python tools/sps_atomic_wait.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "dellyhip.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "--cuda-device-only", "-S",
                       "-o", out, os.path.join(ROOT, "delly_amd", "csrc", "dellyhip.hip")] + sys.argv[1:], cwd=os.path.join(ROOT, "delly_amd", "csrc"), stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN2dh19split_sparse_kernelENS_9SplitArgsE:")][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
body = [l for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((";", "."))]
for a in [i for i, l in enumerate(body) if "_atomic_add" in l and "sc0" in l]:   # (sc0: the returning form -- the work counter)
    w = next((i for i in range(a + 1, len(body)) if "s_waitcnt" in body[i] and "vmcnt" in body[i]), None)
    nxt = body[w + 1].strip() if w is not None and w + 1 < len(body) else ""
    print("atomic at instruction %d: first vmcnt wait %s instructions later, followed by `%s`" % (a, "none" if w is None else w - a, nxt))
print("(the atomic inside the junction loop should be followed by its wait only at the ds_write that parks it, ~1 500 instructions later;"
      " the one at the loop top -- a junction that left early -- is used at once)")
