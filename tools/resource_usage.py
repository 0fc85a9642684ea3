"""registers / LDS / scratch / occupancy of every kernel of libdellyhip.so, from the device assembly of the SAME sources and
flags the library is built with (delly_amd/build.py); needs hipcc only, no GPU:  python tools/resource_usage.py > profiles/r04/resource_usage.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "dellyhip.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "--cuda-device-only", "-S",
                       "-o", out, os.path.join(ROOT, "delly_amd", "csrc", "dellyhip.hip")], cwd=os.path.join(ROOT, "delly_amd", "csrc"), stderr=subprocess.DEVNULL)
name, rows, cur = None, [], {}
for ln in open(out):
    m = re.match(r"^(_Z\S+|[a-z_0-9]+_kernel\S*):\s", ln)
    if m:
        name = m.group(1)
    for key, tag in (("; NumVgprs:", "vgpr"), ("; NumAgprs:", "agpr"), ("; ScratchSize:", "scratch"), ("; Occupancy:", "occ"), ("; LDSByteSize:", "lds"), ("; codeLenInByte =", "code")):
        if ln.startswith(key):
            cur[tag] = int(ln[len(key):].split()[0])
    if ln.startswith("; Occupancy:") and name:
        rows.append((name, dict(cur)))
        cur = {}
for name, r in rows:
    print("%-86s vgpr %3d agpr %3d scratch %4d occ %d lds %6d code %6d" % (name[:86], r.get("vgpr", 0), r.get("agpr", 0), r.get("scratch", 0), r.get("occ", 0), r.get("lds", 0), r.get("code", 0)))
