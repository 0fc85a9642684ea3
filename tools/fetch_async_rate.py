"""What the asynchronous return (dellyhip_batch_fetch_begin / _end) costs on an idle device: 10 000 C2 junctions, the batch finished,
LAPS returns one after the other; beside it the blocking dellyhip_batch_fetch into the same pinned memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import abi, refine, synth

LAPS = 30
ctx = refine.Context()
b = synth.make_batch(10000, mode="c2")
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
RB = abi.result_dtype().itemsize
rec = np.zeros(10000 * RB + 64, dtype=np.uint8)
blob = np.zeros(10000 * 1400 + (1 << 20), dtype=np.uint8)
ctx.host_register(rec.ctypes.data, rec.nbytes)
ctx.host_register(blob.ctypes.data, blob.nbytes)
for name, fn in (("blocking fetch_into", lambda: rb.fetch_into(rec, blob)),
                 ("fetch_begin + fetch_end", lambda: (rb.fetch_begin(rec, blob), rb.fetch_end())[1])):
    used = fn()
    t0 = time.perf_counter()
    for _ in range(LAPS):
        used = fn()
    dt = (time.perf_counter() - t0) / LAPS
    total = used + 10000 * RB
    print("%-26s %.3f ms per return, %.1f MB -> %.1f GB/s" % (name, dt * 1e3, total / 1e6, total / dt / 1e9), flush=True)
rb.free()
ctx.close()
