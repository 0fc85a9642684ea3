"""Per-phase cost of msa_kernel (one junction per wavefront) from a -DDH_MSA_TIMING build:
  python delly_amd/build.py  -> side file:  python -c "from delly_amd import build; build.build_lib(out='/tmp/libmsa_t.so', extra_flags=['-DDH_MSA_TIMING'])"
  DELLYHIP_LIB=/tmp/libmsa_t.so DELLYHIP_MSA_ONLY=1 python tools/msa_phases.py [n_junctions] [n_reads]
Times are wall clock of the junction's wavefront (10 ns ticks) with the chip as full as the batch makes it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = synth.make_batch(n, mode="c2", n_reads=nr)
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
rb.run(); rb.sync()
r, _ = rb.fetch()
tot = r["matches"].astype(np.float64)
print("junctions %d, reads %d; wavefront time per junction: mean %.1f us, median %.1f, p99 %.1f" % (n, nr, tot.mean() / 100, np.median(tot) / 100, np.percentile(tot, 99) / 100))
names = (("c_start", "reads + match masks + all-pairs LCS"), ("c_end", "UPGMA"), ("r_start", "column types + score tables"),
         ("r_end", "Gotoh DP + traceback"), ("hom_left", "_createAlignment"), ("hom_right", "consensus"))
acc = 0.0
for f, name in names:
    v = r[f].astype(np.float64)
    acc += v.mean()
    print("  %-40s %7.1f us  %5.1f %%" % (name, v.mean() / 100, 100 * v.mean() / tot.mean()))
print("  %-40s %7.1f us  %5.1f %%" % ("(rest: setup, node bookkeeping)", (tot.mean() - acc) / 100, 100 * (tot.mean() - acc) / tot.mean()))
cells = r["mismatches"].astype(np.float64)
steps = r["cons_left"].astype(np.float64)
print("DP cells per junction %.0f; row-steps issued %.0f x 64 lanes = %.0f lane-cells (%.0f %% useful); leaf x leaf merges %.1f of %d"
      % (cells.mean(), steps.mean(), steps.mean() * 64, 100 * cells.mean() / (steps.mean() * 64), r["score_best"].mean(), nr - 1))
print("  of the first line: read offsets / lengths %.1f us, match masks + matrix init %.1f us, all-pairs LCS %.1f us" % (r["sv_start"].mean() / 100, r["sv_end"].mean() / 100, r["ins_len"].mean() / 100))
print("paired passes per junction %.2f; their tracebacks %.1f us per junction (inside 'Gotoh DP + traceback')" % (r["ci_wiggle"].mean(), r["hom_len"].mean() / 100))
t0 = r["ref_left"].astype(np.int64)
span = (t0.max() - t0.min()) / 100.0
print("first..last junction start %.0f us" % span)
# the slowest junctions: what kind, when did they start, which phase
order = np.argsort(-tot)[:8]
for j in order:
    print("  slow junction %5d kind %-8s start %6.0f us total %6.0f us: lcs %5.0f upgma %4.0f tables %4.0f dp %5.0f emit %4.0f; paired passes %d, leaf merges %d, cells %d, row-steps %d" % (
        j, b.truth[j]["kind"], (t0[j] - t0.min()) / 100.0, tot[j] / 100, r["c_start"][j] / 100, r["c_end"][j] / 100, r["r_start"][j] / 100, r["r_end"][j] / 100,
        r["hom_left"][j] / 100, r["ci_wiggle"][j], r["score_best"][j], r["mismatches"][j], r["cons_left"][j]))
q = np.percentile(tot, [10, 50, 90, 99, 99.9]) / 100
print("percentiles of the per-junction wavefront time: p10 %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f us" % tuple(q))
late = t0 - t0.min() > 0.8 * (t0.max() - t0.min())
print("junctions started in the last fifth of the launch: %d, mean time %.0f us (all: %.0f)" % (late.sum(), tot[late].mean() / 100, tot.mean() / 100))
