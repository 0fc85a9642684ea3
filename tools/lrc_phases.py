"""Where the long-read consensus kernels spend a junction's time (CHANGELOG.md 3.8 / 3.9): a -DDH_LR_TIMING build of the library
(tools/bin/libdellyhip_lrt.so, built on the CPU box: `python tools/lrc_phases.py --build`) sums per-phase wall-clock ticks over a
launch (lrmsa_kernel.hpp: dh_lrt).  On the GPU box: DELLYHIP_LIB=tools/bin/libdellyhip_lrt.so python tools/lrc_phases.py [rows]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libdellyhip_lrt.so")
SLOTS = {1: "seeding (k-mer tables, diagonal votes)", 2: "superstring NW paths (whole; parts in 7-9, 12)", 3: "buildSuperstring", 4: "column votes",
         5: "forward location pass", 6: "reverse location pass", 7: "Hirschberg last-row passes", 8: "direction fill of base rectangles",
         9: "tracebacks + op reversal", 10: "convertAlignment", 11: "final consensus + trimming",
         13: "progressive NW / HW paths (whole; parts in 5-9)"}   # (slot 12 only re-arms the lap clock in front of a phase: not a phase)

if "--build" in sys.argv:
    from delly_amd import build
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    build.build_lib(out=LIB, extra_flags=["-DDH_LR_TIMING"], verbose=True)
    sys.exit(0)

os.environ.setdefault("DELLYHIP_LIB", LIB)
from delly_amd import abi, refine, synth  # noqa: E402

rows = [a for a in sys.argv[1:] if not a.startswith("-")] or ["lrins:512", "lr:768"]
for row in rows:
    mode, n = row.split(":")
    n = int(n)
    b = synth.make_batch(n, mode=mode, n_reads=15, sub_rate=0.06)
    ctx = refine.Context(params=abi.params_lr(realign=True))
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync()
    out = (C.c_uint64 * 32)()
    ctx.lib.dellyhip_debug_lrt(out, 32)     # (clears)
    t0 = time.perf_counter()
    rb.run(); rb.sync()
    dt = time.perf_counter() - t0
    ctx.lib.dellyhip_debug_lrt(out, 32)
    nj = max(1, int(out[14]))
    print("%s x %d: step %.1f ms; %d junctions through the consensus kernel, %.2f ms per junction wavefront" % (mode, n, dt * 1e3, nj, out[15] / nj / 1e5))
    for k in sorted(SLOTS):
        print("   %5.2f ms  %4.1f %%  %s" % (out[k] / nj / 1e5, 100.0 * out[k] / max(1, out[15]), SLOTS[k]))
    if out[20]:
        print("   banded pair distances: %d pairs, %d not certified by the band (full pass)" % (out[20], out[21]))
    rb.free()
    ctx.close()
