import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from sparse_needle import dense, sparse

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(a, rate):
    a = a.copy()
    for i in range(a.size):
        u = rng.random()
        if u < rate:
            a[i] = ACGT[rng.integers(0, 4)]
    return a


def case(kind):
    n = int(rng.integers(60, 260))
    alpha = ACGT if kind != "lowcx" else ACGT[:2]
    G = alpha[rng.integers(0, alpha.size, n)]
    if kind == "repeat":
        unit = ACGT[rng.integers(0, 4, int(rng.integers(2, 9)))]
        G = np.tile(unit, n // unit.size + 1)[:n].copy()
        for i in rng.integers(0, n, 6):
            G[i] = ACGT[rng.integers(0, 4)]
    m = int(rng.integers(20, 70))
    if kind == "noref":
        a = int(rng.integers(0, n - m))
        cons = G[a:a + m]
    elif kind == "junk":
        cons = ACGT[rng.integers(0, 4, m)]
    else:
        a = int(rng.integers(0, n // 2 - m // 2)) if n // 2 - m // 2 > 0 else 0
        b = int(rng.integers(n // 2, max(n // 2 + 1, n - m // 2)))
        cons = np.concatenate([G[a:a + m // 2], G[b:b + (m - m // 2)]])
    rate = float(rng.choice([0.0, 0.01, 0.03, 0.08]))
    cons = mutate(cons, rate)
    if rng.random() < 0.2 and cons.size > 8:       # an indel in the consensus
        p = int(rng.integers(2, cons.size - 2))
        cons = np.delete(cons, p) if rng.random() < 0.5 else np.insert(cons, p, ACGT[rng.integers(0, 4)])
    if rng.random() < 0.1:
        G = G.copy(); G[rng.integers(0, n, 3)] = ord('N')
    return cons.tobytes(), G.tobytes()


stats = dict(n=0, resolved=0, found=0)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
    kind = ["del", "del", "del", "repeat", "lowcx", "noref", "junk"][it % 7]
    cons, ref = case(kind)
    d = dense(cons, ref)
    for s in (2, 4, 8, 16):
        sp = sparse(cons, ref, s)
        if sp is None:
            continue
        stats["resolved"] += s == 8
        keys = ["found"]
        if sp["found"]:
            keys += ["best", "consLeft", "refLeft", "refRight", "opsF", "opsR"]
        for k in keys:
            assert sp[k] == d.get(k), (it, kind, s, k, sp[k], d.get(k), cons, ref)
        if sp["unsplit"] is not None:
            assert sp["unsplit"] == d["unsplit"], (it, kind, s, "unsplit")
        if not sp["found"]:
            assert d["found"] is False, (it, kind, s)
    stats["n"] += 1
    stats["found"] += bool(d["found"])
print(stats)
