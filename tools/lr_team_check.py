"""lr_dense_team_kernel against the strips on lr_kernel's own wavefronts (DELLYHIP_LR_TEAMS=0): records bit for bit, kernel time, how
many junctions the teams took.  python tools/lr_team_check.py [n_junctions]"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from delly_amd import refine, synth, abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = abi.params_lr(realign=True)
b = synth.make_batch(n, mode="lr", sub_rate=0.01)
out = {}
for teams in ("0", os.environ.get("TEAMS", "64")):
    os.environ["DELLYHIP_LR_TEAMS"] = teams
    ctx = refine.Context(params=P)
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    t0 = time.perf_counter()
    rb.run()
    print("teams", teams, "launched", flush=True)
    rb.sync()
    print("teams", teams, "synced", flush=True)
    print("teams", teams, "first run %.1f ms" % ((time.perf_counter() - t0) * 1e3), "stats", rb.lr_team_stats(), flush=True)
    t0 = time.perf_counter()
    for _ in range(3): rb.run()
    rb.sync()
    print("teams", teams, "ms/step %.2f" % ((time.perf_counter() - t0) / 3 * 1e3), "stats", rb.lr_team_stats(), flush=True)
    r, blob = rb.fetch()
    out[teams] = (r.copy(), bytes(blob))
    rb.free(); ctx.close()
(a, ba), (c, bc) = out["0"], out[os.environ.get("TEAMS", "64")]
same = all(np.array_equal(a[f], c[f]) for f in a.dtype.names) and ba == bc   # (the records' padding word is not part of the ABI)
print("records (every field) and blob identical:", same, flush=True)
if not same:
    print("  records equal", a.tobytes() == c.tobytes(), "blob equal", ba == bc, "blob sizes", len(ba), len(bc))
    if ba != bc and len(ba) == len(bc):
        d = np.nonzero(np.frombuffer(ba, np.uint8) != np.frombuffer(bc, np.uint8))[0]
        print("  blob differs at", d[:10], "count", d.size, "cons_off of some", a["cons_off"][:3], a["allele_off"][:3])
        which = np.searchsorted(a["cons_off"], d[0], side="right") - 1
        print("  first differing junction", which, "status", a["status"][which], c["status"][which], "ok", a["ok"][which], c["ok"][which])
    ra, rc = np.frombuffer(a.tobytes(), np.uint8).reshape(a.size, -1), np.frombuffer(c.tobytes(), np.uint8).reshape(a.size, -1)
    rows, cols = np.nonzero(ra != rc)
    print("  differing record bytes: junctions", sorted(set(rows.tolist()))[:8], "byte offsets", sorted(set(cols.tolist()))[:16], "of", ra.shape[1], "; field offsets", {f: a.dtype.fields[f][1] for f in a.dtype.names})
    for f in a.dtype.names:
        d = np.nonzero(a[f] != c[f])[0]
        if d.size: print("  field", f, "differs at", d[:8], a[f][d[:4]], c[f][d[:4]])
