"""the sparse kernel per SV type: mode "mixed" split by kind (unit U, consensus given)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
b = synth.make_batch(6000, mode="mixed")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
kinds = sorted(set(t["kind"] for t in b.truth))
def rate(sub, label):
    rb = ctx.upload(sub)
    rb.run(); rb.sync(); rb.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(5):
        rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / 5
    res, _ = rb.fetch()
    ran = (res["status"] == 0) & (res["score_best"] != -1)
    deficit = (res["cons_len"] - res["score_best"])[ran]
    print("%-10s n %5d: %.3f ms per step, sparse kernel %.3f ms, left %d, ok %d, cons_len %d..%d, ref_len %d..%d, deficit mean %.1f max %d" % (
        label, sub.n, dt * 1e3, rb.dp_kernel_ms(), rb.sparse_left(), int(res["ok"].sum()), res["cons_len"].min(), res["cons_len"].max(), res["ref_len"].min(), res["ref_len"].max(),
        deficit.mean() if deficit.size else -1, deficit.max() if deficit.size else -1), flush=True)
    rb.free()
rate(b, "all")
for k in kinds:
    idx = np.array([i for i, t in enumerate(b.truth) if t["kind"] == k])
    for sel in sorted(set(int(b.junctions["svid"][i]) % 12 for i in idx)):
        ii = np.array([i for i in idx if int(b.junctions["svid"][i]) % 12 == sel])
        rate(synth.subset(b, ii), "%s/%d" % (k, sel))
