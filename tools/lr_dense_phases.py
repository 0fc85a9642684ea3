"""the dense strip fallback of lr_kernel on the lr_c4_align_consensus batch: which junctions take it, where its time goes (needs a
library built with -DDH_LR_TIMING: DELLYHIP_LIB=tools/bin/lib_lrtiming.bin), and the kernel time of the batch without them"""
import sys, time, collections
sys.path.insert(0, '/root/repo')
import numpy as np
from delly_amd import refine, synth, abi
import bench
P = abi.params_lr(realign=True)
b = synth.make_batch(2048, mode="lr", sub_rate=0.01)
ctx = refine.Context(params=P)
ctx.set_chromosomes(b.chroms)
def run(bb, tag):
    rb = ctx.upload(bb)
    rb.run(); rb.sync()
    t0 = time.perf_counter()
    for _ in range(3): rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / 3
    r, _ = rb.fetch()
    print(tag, "junctions", bb.n, "ms/step %.2f" % (dt * 1e3))
    rb.free()
    return r
r = run(b, "all")
kinds = np.array([t["kind"] for t in b.truth])
deficit = r["cons_len"] - r["score_best"]
dense = (r["score_unsplit"] != -(1 << 30)) & (deficit > 300) & (r["status"] == 0)
print("dense", int(dense.sum()), collections.Counter(kinds[dense]), "ok among them", int(r["ok"][dense].sum()))
t = r["reserved"].astype(np.uint32)[dense]
for name, s in (("setup+sparse attempt", 0), ("R strips", 8), ("M strips", 16), ("winner+dir+traces+masks", 24)):
    v = ((t >> s) & 255) * 0.2
    print("  %-26s median %.1f ms  max %.1f ms" % (name, np.median(v), v.max()))
print("  cons_len", np.percentile(r["cons_len"][dense], [0, 50, 100]), "ref_len", np.percentile(r["ref_len"][dense], [0, 50, 100]))
keep = np.nonzero(~dense)[0]
sub = synth.Batch(b.chroms, b.junctions[keep].copy(), b.seq_blob, b.seq_off, b.with_msa, [b.truth[i] for i in keep])
run(sub, "without the dense junctions")
print("lib", getattr(ctx.lib, "_name", None), "reserved nonzero", int((r["reserved"] != 0).sum()), "of", r.size, "dense reserved", r["reserved"][dense][:8], "status", collections.Counter(r["status"].tolist()))
tt = r["reserved"].astype(np.uint32)[~dense & (r["ok"] == 1)]
print("sparse junctions x50us: orient %d sparse %d masks %d detect %d" % tuple(np.median((tt >> s) & 255) for s in (0, 8, 16, 24)), "max", [int(((tt >> s) & 255).max()) for s in (0, 8, 16, 24)])
if "--spans" in sys.argv:   # library built with -DDH_LR_TIMING=2
    v = r["reserved"].astype(np.uint32)
    st, en = (v >> 16).astype(np.int64), (v & 0xffff).astype(np.int64)
    t0 = st.min()
    st, en = (st - t0) & 0xffff, (en - t0) & 0xffff
    print("junction spans (ms): start percentiles", np.percentile(st, [0, 50, 90, 100]) * 0.05, "end percentiles", np.percentile(en, [10, 50, 90, 99, 100]) * 0.05)
    dur = (en - st) * 0.05
    print("duration ms: sparse median %.1f p90 %.1f max %.1f | dense median %.1f max %.1f" % (np.median(dur[~dense]), np.percentile(dur[~dense], 90), dur[~dense].max(), np.median(dur[dense]), dur[dense].max()))
    late = np.argsort(en)[-8:]
    print("last to finish: end", en[late] * 0.05, "start", st[late] * 0.05, "dense", dense[late], "deficit", deficit[late], "m", r["cons_len"][late], "n", r["ref_len"][late])
if "--spans" in sys.argv:
    slow = np.argsort(np.where(dense, 0, dur))[-12:]
    print("slowest sparse junctions:", [(int(i), round(float(dur[i]), 1), int(deficit[i]), int(r["ok"][i]), str(kinds[i]), int(r["cons_len"][i]), int(r["ref_len"][i])) for i in slow])
    np.save("gpurun_out/lr_slow_idx.npy", slow)
    print("duration by deficit bucket:", [(lo, round(float(np.median(dur[(~dense) & (deficit >= lo) & (deficit < lo + 16)])), 1), int(((~dense) & (deficit >= lo) & (deficit < lo + 16)).sum())) for lo in range(0, 112, 16) if ((~dense) & (deficit >= lo) & (deficit < lo + 16)).any()])
if "--idx" in sys.argv:
    idx = [int(x) for x in sys.argv[sys.argv.index("--idx") + 1].split(",")]
    for i in idx:
        v = int(np.uint32(r["reserved"][i]))
        print(i, "bytes x50us", [(v >> s) & 255 for s in (0, 8, 16, 24)], "deficit", int(deficit[i]), "ok", int(r["ok"][i]))
