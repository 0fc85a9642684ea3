import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from delly_amd import refine, synth
import os
b = synth.make_batch(int(os.environ.get("N", "10000")), mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
rb.run(); rb.sync()
r, _ = rb.fetch()
t0 = r["c_start"].astype(np.int64); t1 = r["c_end"].astype(np.int64)
sel = (t1 > t0)
t0 = t0[sel]; t1 = t1[sel]
base = t0.min()
print("junctions timed", sel.sum(), "span %.0f us" % ((t1.max() - base) / 100), "sum of junction times %.1f ms" % ((t1 - t0).sum() / 1e5))
print("per junction us: mean %.0f median %.0f p99 %.0f max %.0f" % tuple(x / 100 for x in ((t1 - t0).mean(), np.median(t1 - t0), np.percentile(t1 - t0, 99), (t1 - t0).max())))
for f, name in (("r_start", "levels"), ("r_end", "last eval+traces"), ("hom_left", "masks+detect")):
    v = r[f][sel]; print(name, "mean %.0f max %.0f" % (v.mean(), v.max()))
dtj = (t1 - t0) / 100.0
for S in (0, 2, 4, 6, 8, 16):
    ss = r["hom_right"][sel] == S
    if ss.any(): print("S", S, "n", ss.sum(), "total %.0f levels %.0f eval %.0f post %.0f setup %.0f" % (dtj[ss].mean(), r["r_start"][sel][ss].mean(), r["r_end"][sel][ss].mean(), r["hom_left"][sel][ss].mean(), (dtj[ss] - r["r_start"][sel][ss] - r["r_end"][sel][ss] - r["hom_left"][sel][ss]).mean()))
# concurrency over time
edges = np.linspace(base, t1.max(), 21)
for a, bb in zip(edges[:-1], edges[1:]):
    mid = (a + bb) / 2
    print("t=%5.0f us active %d" % ((mid - base) / 100, ((t0 <= mid) & (t1 > mid)).sum()))
print("starts in first 50us:", (t0 < base + 5000).sum())
kinds = np.array([t["kind"] for t in b.truth])
dt = (r["c_end"].astype(np.int64) - r["c_start"].astype(np.int64)) / 100
order = np.argsort(-dt)[:12]
for j in order:
    print(j, kinds[j], "S", r["hom_right"][j], "us %.0f levels %d eval %d post %d ok %d" % (dt[j], r["r_start"][j], r["r_end"][j], r["hom_left"][j], r["ok"][j]))
import collections
slow = dt > 400
print(collections.Counter(kinds[slow]), collections.Counter(kinds))
