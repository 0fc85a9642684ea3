"""CPU prototype of the SPARSE longNeedle (exact): furthest-reaching tables over (deficit level, diagonal) instead of the
dense (m+1) x (n+1) matrices of src/needle.h:45-222.  Checked here against a dense restatement of the reference's
procedure (itself checked against oracle/ port rows) on random and adversarial inputs before any HIP is written.

Deficit D[r][c] = r - M[r][c] >= 0 (M = the forward matrix, +1/-1/-1, horizontal gaps free in rows 0 and m):
  D[0][c] = 0, D[r][0] = 2r, interior: D = min(diag + (0 | 2), up + 2, left + 1); row m: left + 0.
A cell matters for the split only if D <= s where s >= m - bestScore: every cell on an optimal split path, every cell
the reference's traceback compares against successfully, the join winner and refRight all have D_F + D_R <= m - best.
FR[d][k] = last row r on diagonal k (= c - r) with D_int[r][c] <= d  (D_int: row m treated as interior).
"""
import numpy as np

COMP = {ord('A'): ord('T'), ord('C'): ord('G'), ord('G'): ord('C'), ord('T'): ord('A'), ord('N'): ord('N')}


def revcomp_ref(s):
    """reverseComplement of src/util.h:549-563 (quirk: a non-ACGTN reversed letter keeps the ORIGINAL byte at that index)"""
    n = len(s)
    out = bytearray(s)
    for i in range(n):
        ch = s[n - 1 - i]
        if 97 <= ch <= 122:
            ch -= 32
        if ch in COMP:
            out[i] = COMP[ch]
    return bytes(out)


def dense(cons, ref):
    """the reference's procedure, dense.  -> dict(found, unsplit, revmn, best, consLeft, refLeft, refRight, opsF, opsR)"""
    m, n = len(cons), len(ref)

    def fill(a, b):
        M = np.zeros((m + 1, n + 1), dtype=np.int32)
        for r in range(1, m + 1):
            M[r][0] = M[r - 1][0] - 1
        for r in range(1, m + 1):
            hg = 0 if r == m else -1
            ar = a[r - 1]
            row, prev = M[r], M[r - 1]
            for c in range(1, n + 1):
                x = prev[c - 1] + (1 if ar == b[c - 1] else -1)
                y = prev[c] - 1
                z = row[c - 1] + hg
                row[c] = max(x, y, z)
        return M
    M = fill(cons, ref)
    rc, rr = revcomp_ref(cons), revcomp_ref(ref)
    R = fill(rc, rr)
    out = dict(found=False, unsplit=int(M[m][n]), revmn=int(R[m][n]))
    if M[m][n] != R[m][n]:
        return out
    BM = np.maximum.accumulate(M, axis=1)
    BR = np.maximum.accumulate(R, axis=1)
    best = int(M[m][n])
    cl = rl = 0
    for r in range(m + 1):
        tot = BM[r] + BR[m - r][::-1]
        c = int(np.argmax(tot))
        if tot[c] > best:
            best, cl, rl = int(tot[c]), r, c
    cr = m - cl
    rright = 0
    for t in range(0, n - rl + 1):
        if M[cl][rl] + R[cr][t] == best:
            rright = t
    out.update(best=best, consLeft=cl, refLeft=rl, refRight=rright)
    if best == M[m][n]:
        return out

    def trace(X, r, c):
        ops = []
        while r > 0 or c > 0:
            hg = 0 if (r == 0 or r == m) else -1
            if r > 0 and X[r][c] == X[r - 1][c] - 1:
                ops.append('v'); r -= 1
            elif c > 0 and X[r][c] == X[r][c - 1] + hg:
                ops.append('h'); c -= 1
            else:
                ops.append('s'); r -= 1; c -= 1
        return ''.join(ops)
    out.update(found=True, opsF=trace(M, cl, rl), opsR=trace(R, cr, rright))
    return out


NEG = -(1 << 20)


def fr_tables(a, b, s):
    """FR[d][k + m] for d = 0..s, diagonals k = -m..n (k = col - row); NEG = no cell of cost <= d on that diagonal"""
    m, n = len(a), len(b)
    ND = n + m + 1
    FR = np.full((s + 1, ND), NEG, dtype=np.int64)

    def extend(r, k):
        c = r + k
        while r < m and c < n and a[r] == b[c]:
            r += 1
            c += 1
        return r
    for d in range(s + 1):
        for k in range(-m, n + 1):
            best = NEG
            if d == 0:
                if k >= 0:
                    best = 0                       # row 0, column k: D = 0
            else:
                best = FR[d - 1][k + m]            # cost <= d - 1 is cost <= d
                if k - 1 >= -m:                    # horizontal move (r, c-1) -> (r, c): +1, interior rows only (row 0 costs 0 but is 0 anyway)
                    r = FR[d - 1][k - 1 + m]
                    if r >= 0 and r + k <= n and r + k >= 1:
                        best = max(best, r)
                if d >= 2:
                    r = FR[d - 2][k + m]           # mismatch (r, c) -> (r+1, c+1)
                    if r >= 0 and r + 1 <= m and r + 1 + k <= n and r + k >= 0:
                        best = max(best, r + 1)
                    if k + 1 <= n:                 # vertical move (r, c) on diagonal k+1 -> (r+1, c) on k: +2
                        r = FR[d - 2][k + 1 + m]
                        if r >= 0 and r + 1 <= m and r + 1 + k >= 0:
                            best = max(best, r + 1)
                if k < 0 and 2 * (-k) <= d:        # column 0: D[r][0] = 2r
                    best = max(best, -k)
            if best >= 0:
                best = extend(int(best), k)
            FR[d][k + m] = best
    return FR


def sparse(cons, ref, s):
    """-> the dict of dense() or None when the junction is not resolved at deficit s (go to a larger s / the dense kernels)"""
    m, n = len(cons), len(ref)
    rc, rr = revcomp_ref(cons), revcomp_ref(ref)
    FF = fr_tables(cons, ref, s)
    FRv = fr_tables(rc, rr, s)

    def row_m_deficit(FR):
        for d in range(s + 1):
            if (FR[d] >= m).any():
                return d
        return None
    du, dv = row_m_deficit(FF), row_m_deficit(FRv)

    def first_col_tables(FR):
        # cF[d][r] = first column c with  min_{c' <= c} D[r][c'] <= d  (n + 1 if none); row m: D_int, prefix-min is what the reference's
        # prefix-max of the free-gap row equals
        cF = np.full((s + 1, m + 1), n + 1, dtype=np.int64)
        for d in range(s + 1):
            pm = NEG
            for k in range(-m, n + 1):
                v = FR[d][k + m]
                if v > pm:
                    lo = max(pm + 1, 0)
                    for r in range(lo, int(v) + 1):
                        if r + k >= 0:
                            cF[d][r] = min(cF[d][r], r + k)
                        # (rows of this diagonal left of column 0 do not exist)
                    pm = v
        return cF
    # careful: a diagonal reaching row v covers rows <= v only from its own start row (max(0, -k)) -- rows above the start are not on it
    cF = np.full((s + 1, m + 1), n + 1, dtype=np.int64)
    cR = np.full((s + 1, m + 1), n + 1, dtype=np.int64)
    for tab, FR in ((cF, FF), (cR, FRv)):
        for d in range(s + 1):
            for k in range(-m, n + 1):
                v = int(FR[d][k + m])
                if v < 0:
                    continue
                for r in range(max(0, -k), v + 1):
                    if r + k < tab[d][r]:
                        tab[d][r] = r + k
    # join: smallest total deficit, first row, first column
    bestD, cl, rl, dsel = None, 0, 0, 0
    for r in range(m + 1):
        for tot in range(0, s + 1):
            hit = None
            for d in range(tot, -1, -1):           # largest feasible d gives the smallest column
                e = tot - d
                lo, hi = cF[d][r], n - cR[e][m - r]
                if lo <= n and cR[e][m - r] <= n and lo <= hi:
                    hit = (lo, d)
                    break
            if hit is not None:
                if bestD is None or tot < bestD:
                    bestD, cl, rl, dsel = tot, r, int(hit[0]), hit[1]
                break
    if du is None and bestD is None:
        return None                                 # nothing within s: unresolved
    unsplit = m - du if du is not None else None
    out = dict(found=False, unsplit=unsplit, revmn=(m - dv) if dv is not None else None)
    if bestD is None or (du is not None and bestD >= du):
        if du is None:
            return None
        # no improving split among deficits <= s; splits with deficit > s >= du cannot improve either
        out.update(best=unsplit, consLeft=0, refLeft=0)
        return out
    best = m - bestD
    cr = m - cl
    # D_F at (cl, rl): the prefix-min there
    dF = min(d for d in range(s + 1) if cF[d][cl] <= rl)
    eR = bestD - dF
    # refRight: last t <= n - rl with D_R[cr][t] == eR  <=> FR_R[eR][t - cr] >= cr
    rright = 0
    for t in range(0, n - rl + 1):
        k = t - cr
        if -m <= k <= n and FRv[eR][k + m] >= cr and t >= 0 and (k >= 0 or cr >= -k):
            rright = t
    out.update(best=best, consLeft=cl, refLeft=rl, refRight=rright, found=True)

    def D_at(FR, r, c):
        k = c - r
        for d in range(s + 1):
            if FR[d][k + m] >= r:
                return d
        return None

    def trace(FR, a, b, r, c):
        ops = []
        D = D_at(FR, r, c)
        assert D is not None
        while r > 0 or c > 0:
            if r == 0:
                ops.append('h'); c -= 1
                continue
            k = c - r
            # vertical: D[r-1][c] == D - 2
            if D >= 2 and k + 1 <= n and FR[D - 2][k + 1 + m] >= r - 1 and (k + 1 >= 0 or r - 1 >= -(k + 1)):
                ops.append('v'); r -= 1; D -= 2
            elif c > 0 and D >= 1 and FR[D - 1][k - 1 + m] >= r and (k - 1 >= 0 or r >= -(k - 1)):
                ops.append('h'); c -= 1; D -= 1
            elif c > 0:
                ops.append('s')
                D -= 0 if a[r - 1] == b[c - 1] else 2
                r -= 1; c -= 1
            else:                                   # column 0: only vertical moves remain (D[r][0] = 2r)
                ops.append('v'); r -= 1; D -= 2
        return ''.join(ops)
    out.update(opsF=trace(FF, cons, ref, cl, rl), opsR=trace(FRv, rc, rr, cr, rright))
    return out
