"""Summarises a rocprofv3 --kernel-trace --memory-copy-trace run of tools/bench_stream.py: busy time of the compute queue and
of the copy engines, their overlap, per-kernel totals, copy bandwidths."""
import csv
import glob
import os
import sys

d = sys.argv[1]
kern = [f for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
cop = [f for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)]
ks, cs = [], []
for f in kern:
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in cop:
    for r in csv.DictReader(open(f)):
        cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), int(r.get("Bytes", r.get("Size", 0)) or 0)))
ks.sort(); cs.sort()
if not ks:
    raise SystemExit("no kernel trace found in " + d)
# last 60 % of the run = steady state
t_lo = ks[0][0] + int(0.4 * (ks[-1][1] - ks[0][0]))
ks = [k for k in ks if k[0] >= t_lo]
cs = [c for c in cs if c[0] >= t_lo]
span = max(ks[-1][1], cs[-1][1] if cs else 0) - t_lo


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    out = []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s; out.append((cur_s, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s; out.append((cur_s, cur_e))
    return tot, out


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


kt, ku = union([(s, e) for s, e, _ in ks])
ct, cu = union([(s, e) for s, e, _, _ in cs])
print("steady-state span %.2f ms: kernels busy %.2f ms (%.0f %%), copies busy %.2f ms (%.0f %%), both at once %.2f ms"
      % (span / 1e6, kt / 1e6, 100 * kt / span, ct / 1e6, 100 * ct / span, inter(ku, cu) / 1e6))
by = {}
for s, e, n in ks:
    a = by.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:8]:
    print("  kernel %-60s x%-5d avg %8.1f us" % (n, c, t / c / 1e3))
byd = {}
for s, e, dr, b in cs:
    a = byd.setdefault(dr, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += b
for dr, (c, t, b) in byd.items():
    print("  copy %-20s x%-5d avg %8.1f us  %.1f MB total  %.1f GB/s while copying" % (dr, c, t / c / 1e3, b / 1e6, b / max(t, 1)))
