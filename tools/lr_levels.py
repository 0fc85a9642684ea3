import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from delly_amd import refine, synth, abi
P = abi.params_lr(realign=True)
for name, kw, n in (("given", dict(mode="lr", sub_rate=0.01), 1024), ("msaedlib", dict(mode="lr", n_reads=15, sub_rate=0.06), 768)):
    b = synth.make_batch(n, **kw)
    ctx = refine.Context(params=P)
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync()
    t0 = time.perf_counter()
    for _ in range(3): rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / 3
    ms_split, ms_msa, _ = rb.kernel_ms()
    r, _ = rb.fetch()
    lv = r["reserved"]
    if "--timing" in sys.argv:
        t = r["reserved"].astype(np.uint32)
        t2 = r["reserved"].astype(np.uint32)[r["ok"] == 1]
        print(name, "sparse phases x50us: levels %d tables %d join %d traces %d" % tuple(np.median((t2 >> s) & 255) for s in (0, 8, 16, 24)))
        print(name, "phase medians x50us: orient %d sparse %d masks %d detect %d" % tuple(np.median((t >> s) & 255) for s in (0, 8, 16, 24)), "max", [int(((t >> s) & 255).max()) for s in (0, 8, 16, 24)])
    print(name, "ms/step %.1f split %.1f" % (dt * 1e3, ms_split), "levels hist", np.bincount(np.minimum(lv, 300) // 16)[:20], "deficit m-best", np.percentile((r["cons_len"] - r["score_best"])[r["ok"] == 1], [10, 50, 90, 99]))
    rb.free(); ctx.close()
