import sys, os, time, faulthandler
faulthandler.dump_traceback_later(12, exit=True)
t0 = time.time()
def mark(s): print("[%.2f] %s" % (time.time() - t0, s), flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mark("start")
import numpy as np
from delly_amd import refine, synth
mark("imports")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = synth.make_batch(n, mode="c2")
mark("batch")
ctx = refine.Context()
mark("ctx")
ctx.set_chromosomes(b.chroms)
mark("chrom")
res, blob = ctx.refine(b, want_alignment=False)
mark("done ok=%d best=%s" % (int(res["ok"].sum()), res["score_best"][:4]))
for f in ("score_unsplit", "score_best", "cons_left", "ref_left", "ref_right", "c_start", "c_end", "r_start", "r_end", "hom_left", "hom_right", "matches", "cons_len", "ref_len"):
    print(f, res[f][:8])
