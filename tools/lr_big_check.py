import sys, collections
sys.path.insert(0, "/root/repo")
import numpy as np
from delly_amd import abi, refine, synth
for n in (2048, 4096, 8192):
    b = synth.make_batch(n, mode="lr", sub_rate=0.01)
    ctx = refine.Context(params=abi.params_lr(realign=True))
    ctx.set_chromosomes(b.chroms)
    r, bl = ctx.refine(b)
    kinds = collections.Counter(t["kind"] for t in b.truth)
    okk = collections.Counter(t["kind"] for t, o in zip(b.truth, r["ok"]) if o)
    print(n, "ok", int(r["ok"].sum()), "status", collections.Counter(r["status"].tolist()), "kinds", dict(kinds), "ok by kind", dict(okk))
    idx = np.nonzero(r["ok"] == 0)[0]
    print("  first failing", idx[:10], "last ok", np.nonzero(r["ok"])[0][-5:], "chrom len", [c.size for c in b.chroms])
    ctx.close()
