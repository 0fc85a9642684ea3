"""Regenerates tests/golden/chr18_example.npz: the 200 001 bases of the reference's example chromosome
(/root/reference/example/ref.fa, a piece of human chr18 with its natural homopolymers and (CA)n / (TG)n repeats),
2-bit packed.  The GPU box has no /root/reference, so the parity tests and bench.py's `deficit_sweep` read this file.
Run in the dev container:  python tests/golden/make_real_fixture.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lines = open("/root/reference/example/ref.fa").read().split("\n")
    assert lines[0].startswith(">")
    seq = np.frombuffer("".join(lines[1:]).encode(), dtype=np.uint8)
    code = np.full(256, 255, dtype=np.uint8)
    for k, c in enumerate(b"ACGT"):
        code[c] = k
    c2 = code[seq]
    assert (c2 < 4).all(), "letters outside ACGT: store bytes instead"
    pad = (-c2.size) % 4
    c2 = np.concatenate([c2, np.zeros(pad, dtype=np.uint8)]).reshape(-1, 4)
    packed = (c2[:, 0] | (c2[:, 1] << 2) | (c2[:, 2] << 4) | (c2[:, 3] << 6)).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "chr18_example.npz"), packed=packed, n=np.int64(seq.size),
                        name=np.bytes_(lines[0][1:].encode()))
    print("chr18_example.npz: %d bases" % seq.size)


if __name__ == "__main__":
    main()
