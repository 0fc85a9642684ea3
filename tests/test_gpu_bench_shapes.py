"""-m gpu: bit-compare of the HIP path with the reference itself (oracle/_ref, all host threads) on EXACTLY the
batches bench.py times -- the 10 000-junction C2 headline batch and the side measurements of bench.SIDE_PLAN
(U_full N = 20 / 5, insertions, long-read alignConsensus, long-read msaEdlib + alignConsensus with 15 reads of
2.2 kb at 6 % error).  The long-read CPU legs are bounded to what the reference finishes in seconds
(>= 256 / >= 64 junctions, the prefix of the benched batch: synth batches are counter-based, junction j does
not depend on the batch size)."""
import os
import sys

import numpy as np
import pytest

from delly_amd import abi, refine, synth
from util import CORE, compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (SIDE_PLAN only; main() is not run)

THREADS = os.cpu_count() or 1
# junctions compared per workload (None = the whole benched batch)
COMPARE_N = {"u_full_n20": None, "u_full_n20_10k_junctions": 1000, "u_full_n5": None, "ins_svt4": None, "lr_c4_align_consensus": 256,
             "lr_c4_msaedlib_n15": 64, "lr_ins_msawfa_n15": 64}


def _check(ctx, ref, b, params, label, n_cmp=None):
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=False)
    sub = b if n_cmp is None or n_cmp >= b.n else bench._subbatch(b, n_cmp)
    rr, rb = ref.refine_batch(sub, want_alignment=False, n_threads=THREADS, params=params)
    k = sub.n
    compare(gr[:k], gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label=label)
    return gr


def test_headline_batch_10000_c2_junctions_vs_reference(gpu_ctx, reference):
    b = synth.make_batch(10000, mode="c2")
    gr = _check(gpu_ctx, reference, b, None, "bench headline (10 000 C2)")
    assert int(gr["ok"].sum()) == 9900   # bench.py's refined_ok


@pytest.mark.parametrize("name", [x[0] for x in bench.SIDE_PLAN if x[0] in COMPARE_N])
def test_side_measurement_batches_vs_reference(reference, name):
    _, n, _, kw = [x for x in bench.SIDE_PLAN if x[0] == name][0]
    lr = kw["mode"].startswith("lr")
    params = abi.params_lr(realign=True) if lr else abi.params_sr()
    ctx = refine.Context(params=params)
    try:
        b = synth.make_batch(n, **kw)
        gr = _check(ctx, reference, b, params, name, COMPARE_N[name])
        assert int((gr["status"] != 0).sum()) == 0
        assert int(gr["ok"].sum()) > 0.75 * n
    finally:
        ctx.close()
