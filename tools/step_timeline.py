"""One stream's kernels over a few steps of the headline arrangement, from a rocprofv3 kernel trace: start, duration and the gap in
front of every kernel -- what a step boundary costs (CHANGELOG.md round 6, "Between the kernels").  On the GPU box:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $REPO/bench.py --steps 20 --warmup 3 \\
        --repeats 2 --no-alone --no-cpu-baseline --no-extras --no-host-inclusive
    python $REPO/tools/step_timeline.py /tmp/tl"""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tl"
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sp = [r for r in rows if "split_sparse_kernel" in r["Kernel_Name"]]
print("kernels", len(rows), "sparse launches", len(sp))
lo = int(sp[-17]["Start_Timestamp"])   # a window of steady state: the last 16 sparse launches
byq = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) >= lo:
        byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:2]:
    print("queue", q)
    prev = None
    for r in rs[:12]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("dh::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:26]
        print("   %-26s start %8.1f us  dur %7.1f  gap in front %6.1f" % (name, (s - lo) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
        prev = e
starts = [int(r["Start_Timestamp"]) for r in sp[-17:]]
print("sparse kernel, start to start (us):", [round((b - a) / 1e3, 1) for a, b in zip(starts, starts[1:])])
