"""profiling build (-DDH_SPS_DBG): where the level loop of split_sparse_kernel spends its wall time, summed over the wavefronts
of one 10 000-junction launch.  DELLYHIP_LIB=tools/bin/lib_dbg.bin python tools/sps_dbg.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
b = synth.make_batch(10000, mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
out = (C.c_uint64 * 16)()
ctx.lib.dellyhip_debug_read(out, 16)
rb.run(); rb.sync()
ctx.lib.dellyhip_debug_read(out, 16)
v = list(out)
calls, levels = v[0], v[1]
us = lambda t: t / 100.0
print("level-block calls %d, levels %d (%.2f per junction)" % (calls, levels, levels / 1e4))
print("per level: diagonal loop %.2f us, tail (reductions + barrier) %.2f us" % (us(v[2]) / levels, us(v[3]) / levels))
print("per call : epilogue (wait for the spill stores) %.2f us, prologue (tile zeroing / barrier) %.2f us" % (us(v[4]) / calls, us(v[7]) / calls))
print("extensions: %d calls (%.2f per level), %.2f us each" % (v[5], v[5] / levels, us(v[6]) / max(v[5], 1)))
print("per junction: loop %.1f tail %.1f epilogue %.1f prologue %.1f us" % (us(v[2]) / 1e4, us(v[3]) / 1e4, us(v[4]) / 1e4, us(v[7]) / 1e4))
