import sys
sys.path.insert(0, '/root/repo')
from delly_amd import refine, synth, abi
b = synth.make_batch(512, mode="lrins", n_reads=15, sub_rate=0.06)
ctx = refine.Context(params=abi.params_lr(realign=True))
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync()
