import sys, collections
sys.path.insert(0,'/root/repo')
import numpy as np
from delly_amd import refine, synth, abi
P = abi.params_lr(realign=True)
b = synth.make_batch(1024, mode="lr", sub_rate=0.01)
ctx = refine.Context(params=P)
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
r, _ = rb.fetch()
kinds = np.array([t["kind"] for t in b.truth])
lv = r["reserved"]
fb = lv < 16
print("fallback", fb.sum(), collections.Counter(kinds[fb]), "ok among them", r["ok"][fb].sum())
d = r["cons_len"] - r["score_best"]
print("deficit of fallback junctions", np.sort(d[fb])[:30], "cons_len", r["cons_len"][fb][:10], "ref_len", r["ref_len"][fb][:10])
print("all kinds", collections.Counter(kinds))
