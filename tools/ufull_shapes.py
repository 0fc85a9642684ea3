import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from delly_amd import refine, synth
b = synth.make_batch(2000, mode="c2", n_reads=20)
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
r, _ = rb.fetch()
print("cons_len pct", np.percentile(r["cons_len"], [0, 10, 50, 90, 100]), "ref_len pct", np.percentile(r["ref_len"], [0, 50, 100]))
print("m>254:", (r["cons_len"] > 254).sum(), "nd>1408:", ((r["cons_len"] + r["ref_len"] + 1) > 1408).sum())
