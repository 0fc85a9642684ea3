"""Phase clocks of split_sparse_kernel, per deficit level count (a -DDH_LR_TIMING -DDH_SPS_FINE build of the library overwrites
diagnostic fields of the result record with times):
    python -c "from delly_amd import build; build.build_lib(out='tools/bin/libdellyhip_fine.so', extra_flags=['-DDH_LR_TIMING','-DDH_SPS_FINE'])"
    gpurun -- 'DELLYHIP_LIB=tools/bin/libdellyhip_fine.so python tools/sr_fine.py'
Whole-junction view (set-up / levels / evaluation / post, concurrency over time): tools/sr_phases.py on a -DDH_LR_TIMING build.
Where the kernel waits for its work-counter atomic: tools/sps_atomic_wait.py (no GPU needed)."""
import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from delly_amd import refine, synth
b = synth.make_batch(10000, mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
r, _ = rb.fetch()
ok = (r["ok"] == 1)
for S in (0, 2, 4):
    s = ok & (r["hom_right"] == S)
    print("S", S, "n", s.sum(), "us: lists+firstcols %.1f join+refRight %.1f traces %.1f masks %.1f detect %.1f" % tuple(r[f][s].mean() / 10 for f in ("r_start", "r_end", "hom_left", "matches", "mismatches")))

for S in (0, 2):
    s = ok & (r["hom_right"] == S)
    print("S", S, "detect us: findSplit %.1f percentId %.1f homology %.1f coords %.1f alleles+record %.1f" % tuple(r[f][s].mean() / 10 for f in ("score_unsplit", "score_best", "cons_left", "ref_left", "ref_right")))
print("allele_len mean", r["allele_len"][ok].mean(), "hom", r["hom_left"][ok].mean())

kinds = np.array([t["kind"] for t in b.truth])
nr = kinds == "noref"
print("noref n", nr.sum(), "S", np.bincount(r["hom_right"][nr]), "us: lists+firstcols %.1f join %.1f" % (r["r_start"][nr].mean() / 10, r["r_end"][nr].mean() / 10), "max", r["r_start"][nr].max() / 10, r["r_end"][nr].max() / 10)
s = ok & (r["hom_right"] == 0)
print("S 0 level block us: before %.1f clear tile %.1f steps %.1f reductions+end %.1f after %.1f" % tuple(r[f][s].mean() / 10 for f in ("sr_support", "hom_len", "ci_wiggle", "cons_bp", "ins_len")))
