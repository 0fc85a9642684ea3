"""Which source lines the scratch (private-memory) accesses of a kernel come from: the device assembly of dellyhip.hip with line
tables (needs hipcc only, no GPU, ~2 min), every scratch_load / scratch_store attributed to the last .loc in front of it.
  python tools/scratch_sites.py [kernel-name-substring ...] > profiles/r05/scratch_sites.txt
Without arguments: every kernel of profiles' resource table with more than 64 bytes of scratch per lane."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.environ.get("DELLYHIP_ASM")          # reuse an assembly made earlier
if not asm:
    asm = os.path.join(tempfile.mkdtemp(), "dellyhip_g.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "--cuda-device-only",
                           "-S", "-gline-tables-only", "-o", asm, os.path.join(ROOT, "delly_amd", "csrc", "dellyhip.hip")],
                          cwd=os.path.join(ROOT, "delly_amd", "csrc"), stderr=subprocess.DEVNULL)
files, per, size = {}, collections.OrderedDict(), {}
name, loc = None, None
for ln in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
        continue
    m = re.match(r"^(_Z\S+|[a-z_0-9]+_kernel\S*):\s", ln)
    if m:
        name = m.group(1)
        per.setdefault(name, collections.Counter())
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if name and ln.startswith("; ScratchSize:"):
        size[name] = int(ln.split()[2])
    m = re.match(r"\s*scratch_(load|store)", ln)
    if m and name and loc:
        per[name][(loc[0], loc[1], m.group(1))] += 1
want = sys.argv[1:]
for k, c in per.items():
    if (want and not any(w in k for w in want)) or (not want and size.get(k, 0) <= 64):
        continue
    by_line = collections.Counter()
    for (f, l, kind), n in c.items():
        by_line[(f, l)] += n
    print("%s: ScratchSize %d B per lane, %d scratch instructions in the code" % (k, size.get(k, 0), sum(c.values())))
    for (f, l), n in by_line.most_common(8):
        print("    %-22s line %4d: %3d (%d loads, %d stores)" % (f, l, n, c[(f, l, "load")], c[(f, l, "store")]))
