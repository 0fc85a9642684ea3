"""debug helper: HIP vs port on an INS batch, prints the first diverging junctions with splitAlign internals"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyoracle
from delly_amd import synth, refine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = synth.make_batch(n, mode="ins")
P = pyoracle.Oracle("port")
pr, pb = P.refine_batch(b)
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
gr, gb = ctx.refine(b, want_alignment=True)
F = ["ok", "status", "score_unsplit", "score_best", "cons_left", "ref_left", "ref_right", "c_start", "c_end", "r_start", "r_end",
     "hom_left", "hom_right", "matches", "mismatches", "aln_len", "cons_len", "ref_len", "sv_start", "sv_end"]
nb = 0
for i in range(n):
    d = [f for f in F if gr[i][f] != pr[i][f]]
    ga, pa = pyoracle.blob_field(gr[i], gb, "aln"), pyoracle.blob_field(pr[i], pb, "aln")
    if d or ga != pa:
        nb += 1
        if nb <= 6:
            print("junction", i, b.truth[i]["kind"], "diff:", d, "aln_equal", ga == pa)
            print("  hip :", [int(gr[i][f]) for f in F])
            print("  port:", [int(pr[i][f]) for f in F])
            if ga != pa and ga and pa:
                L = len(ga) // 2; M = len(pa) // 2
                print("  hip  ", ga[:L].decode(errors="replace")); print("       ", ga[L:].decode(errors="replace"))
                print("  port ", pa[:M].decode(errors="replace")); print("       ", pa[M:].decode(errors="replace"))
print("mismatching junctions:", nb, "of", n)
