"""-m gpu fuzz parity: perturbed junctions (tests/fuzz.py) under four parameter sets, and chromosomes
cut around the breakpoints -- HIP vs the C restatement on every field of the result record.
The restatement itself is held against the reference build on the same inputs by
tests/test_oracle_golden.py::test_fuzz_port_vs_reference."""
import pytest

import fuzz
from delly_amd import refine
from util import CORE, INTERNAL, compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["c2", "mixed", "ins"])
@pytest.mark.parametrize("pi", [0, 1, 2, 3])
def test_perturbed_junctions(port, mode, pi):
    p = fuzz.params_of(pi)
    b = fuzz.perturbed(240, 5 + pi, mode)
    ctx = refine.Context(params=p)
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=True)
    ctx.close()
    pr, pb = port.refine_batch(b, params=p)
    assert int(pr["ok"].sum()) > 100
    compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="fuzz %s/%d" % (mode, pi))


@pytest.mark.parametrize("mode", ["c2", "mixed", "ins"])
def test_windows_clipped_at_chromosome_ends(port, mode):
    ctx = refine.Context()
    n_ok = 0
    for b in fuzz.clipped(60, 3, mode):
        ctx.set_chromosomes(b.chroms)
        gr, gb = ctx.refine(b, want_alignment=True)
        pr, pb = port.refine_batch(b)
        compare(gr, gb, pr, pb, fields=CORE + INTERNAL, label="clipped " + mode)
        n_ok += int(pr["ok"][0])
    ctx.close()
    assert n_ok > 10
