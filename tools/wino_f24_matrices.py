#!/usr/bin/env python3
"""Cook-Toom matrices of the 1-D minimal filtering algorithm F(2,4) (2 outputs, 4 taps, 5 multiplications) used by
csrc/wino44.hip, derived exactly (fractions) for a choice of interpolation points, checked against the correlation they must
reproduce, and their fp32 error measured on random data at the discriminator conv4 shape (C = 256 accumulations).

    python3 tools/wino_f24_matrices.py"""
from fractions import Fraction as Fr
import itertools
import numpy as np


def solve(M, rhs):
    """least-squares-free exact solve of an over-determined consistent system M x = rhs (fractions)."""
    rows, n = len(M), len(M[0])
    A = [list(M[i]) + [rhs[i]] for i in range(rows)]
    piv = []
    r = 0
    for c in range(n):
        p = next((i for i in range(r, rows) if A[i][c] != 0), None)
        if p is None:
            continue
        A[r], A[p] = A[p], A[r]
        A[r] = [v / A[r][c] for v in A[r]]
        for i in range(rows):
            if i != r and A[i][c] != 0:
                f = A[i][c]
                A[i] = [a - f * b for a, b in zip(A[i], A[r])]
        piv.append(c)
        r += 1
    assert all(all(v == 0 for v in A[i]) for i in range(r, rows)), 'inconsistent'
    x = [Fr(0)] * n
    for i, c in enumerate(piv):
        x[c] = A[i][n]
    return x


def matrices(points, m=2, r=4):
    n = m + r - 1
    a = [Fr(p) for p in points]
    assert len(a) == n - 1
    AT = [[a[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fr(1)
        for l in range(n - 1):
            if l != j:
                f *= a[j] - a[l]
        G.append([a[j] ** k / f for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    Mx = [[AT[i][j] * G[j][k] for j in range(n)] for i in range(m) for k in range(r)]
    BT = [[None] * n for _ in range(n)]
    for p in range(n):
        col = solve(Mx, [Fr(1 if p == i + k else 0) for i in range(m) for k in range(r)])
        for j in range(n):
            BT[j][p] = col[j]
    return AT, G, BT


def check(AT, G, BT, m=2, r=4):
    n = m + r - 1
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(r), rng.standard_normal(n)
    f = lambda M: np.array([[float(v) for v in row] for row in M])
    y = f(AT) @ ((f(G) @ g) * (f(BT) @ d))
    ref = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref), (y, ref)


def fp32_error(AT, G, BT, C=256, trials=4):
    f = lambda M, t: np.array([[float(v) for v in row] for row in M], dtype=t)
    rng = np.random.default_rng(1)
    errs, derr = [], []
    for _ in range(trials):
        g = rng.standard_normal((C, 4, 4)) / np.sqrt(C * 16)
        d = rng.standard_normal((C, 5, 5))
        ref = np.zeros((2, 2))
        for i, j in itertools.product(range(2), range(2)):
            ref[i, j] = (g * d[:, i:i + 4, j:j + 4]).sum()
        for t in (np.float32,):
            U = np.einsum('xa,cab,yb->cxy', f(G, t), g.astype(t), f(G, t)).astype(t)
            V = np.einsum('xa,cab,yb->cxy', f(BT, t), d.astype(t), f(BT, t)).astype(t)
            Mm = (U * V).sum(0, dtype=t)
            y = f(AT, t) @ Mm @ f(AT, t).T
            errs.append(np.abs(y - ref).max() / np.abs(ref).max())
        dd = np.zeros((2, 2), np.float32)
        for i, j in itertools.product(range(2), range(2)):
            dd[i, j] = (g.astype(np.float32) * d.astype(np.float32)[:, i:i + 4, j:j + 4]).sum(dtype=np.float32)
        derr.append(np.abs(dd - ref).max() / np.abs(ref).max())
    return float(np.mean(errs)), float(np.mean(derr))


if __name__ == '__main__':
    for pts in ((0, 1, -1, 2), (0, 1, -1, Fr(1, 2)), (0, 1, -1, -2), (0, Fr(1, 2), Fr(-1, 2), 1), (0, 1, -1, Fr(-1, 2))):
        AT, G, BT = matrices(pts)
        check(AT, G, BT)
        e, de = fp32_error(AT, G, BT)
        print('points', [str(p) for p in pts], ' fp32 rel err winograd %.2e  direct %.2e' % (e, de))
        for nm, M in (('AT', AT), ('G', G), ('BT', BT)):
            print(' ', nm, [[str(v) for v in row] for row in M])
