"""rows measured after a row that released a lot of device memory lose ~0.5 ms per step.  Hypothesis: memory handed
back to the driver (hipFree: the pools' trim, a stream's slots, a context's workspaces -- or the exit of the PREVIOUS process) is
cleared by the driver asynchronously on the GPU, and short launches queue behind / beside that work.  Probe: time the resident
headline step (a) fresh, (b) right after hipFree of GB_FREED GB, (c) 2 s later.   python tools/free_scrub_probe.py [GB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from delly_amd import refine, synth  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
b = synth.make_batch(10000, mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()


def steps(k=200):
    t0 = time.perf_counter()
    for _ in range(k):
        rb.run()
        rb.sync()
    return (time.perf_counter() - t0) / k * 1e3


print("fresh: %.3f ms per step" % steps())
x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda")
x.fill_(1)
torch.cuda.synchronize()
print("with %.0f GB allocated: %.3f ms per step" % (gb, steps()))
del x
t0 = time.perf_counter()
torch.cuda.empty_cache()          # hipFree
print("hipFree of %.0f GB returned after %.1f ms" % (gb, (time.perf_counter() - t0) * 1e3))
for k in range(6):
    print("  %.2f s after the free: %.3f ms per step" % (time.perf_counter() - t0, steps(100)))
time.sleep(2.0)
print("2 s later: %.3f ms per step" % steps())
# the library's own release path: a stream's slots and the pools' trim
st = refine.Stream(ctx, depth=6)
for k in range(12):
    st.submit(b, tag=k)
    if k >= 5:
        st.collect()
while st.pending():
    st.collect()
st.close()
print("after a depth-6 stream was closed: %.3f ms per step" % steps(100))
freed = ctx.trim_memory()
print("dellyhip_trim_memory released %.1f MB; right after: %.3f, %.3f, %.3f ms per step" % (freed / 1e6, steps(100), steps(100), steps(100)))
rb.free()
ctx.close()
