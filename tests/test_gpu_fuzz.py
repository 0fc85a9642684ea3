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


# Without alignment rows the sparse kernel builds the column masks in registers (sparse_masks_regs) and every kernel runs
# _findSplit / the allele cut on register masks (MaskRegs, split_kernel.hpp) -- the runs above ask for the rows and take the LDS
# form.  The last parameter set has no minimum flank and a low identity bar, so that unusual splits survive the filters.
# (A split that starts at the alignment's first column -- cStart or rStart 0: the alleles are then located on the masks with
# select_bit_reg / cnt_before_reg instead of being cut from the coordinates -- does not occur in these batches; that path is
# held to the reference by running this file and the sparse / compact suites on a -DDH_NO_DIRECT_CUT build, which sends EVERY
# junction through it: CHANGELOG.md round 6.)
# Checked against the reference's own code (oracle/_ref), not the restatement: the fifth set is not one the restatement is pinned on.
NO_FLANK = (5, -4, -10, -1, 2, 0, 1000, 100, 0.5, 0)


@pytest.mark.parametrize("mode", ["c2", "mixed", "ins"])
@pytest.mark.parametrize("pi", [0, 1, 2, 3, 4])
def test_perturbed_junctions_without_alignment_rows(reference, mode, pi):
    from delly_amd import abi
    p = abi.Params(*NO_FLANK) if pi == 4 else fuzz.params_of(pi)
    b = fuzz.perturbed(240, 11 + pi, mode)
    ctx = refine.Context(params=p)
    ctx.set_chromosomes(b.chroms)
    gr, gb = ctx.refine(b, want_alignment=False)
    ctx.close()
    rr, rb = reference.refine_batch(b, params=p, want_alignment=False)
    assert int(rr["ok"].sum()) > 60
    compare(gr, gb, rr, rb, fields=CORE, blobs=("cons", "allele"), label="fuzz, no rows %s/%d" % (mode, pi))
