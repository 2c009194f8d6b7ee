"""Emulation of deal.II's default DoF enumeration (DoFHandler::distribute_dofs on a
hyper_cube refined globally `level` times, FE_Q(p), 2-D): cells in Morton order,
per cell first the not-yet-numbered vertices, then lines (left, right, bottom,
top), then the cell interior.  deal.II's PreconditionChebyshev seeds its eigenvalue
estimate with the vector (i % 11) in THIS numbering, so reproducing the printed
estimates of the reference's fixtures needs it."""
import numpy as np


def morton_cells(level):
    cells = [(0, 0)]
    for _ in range(level):
        cells = [(2 * cx + (c & 1), 2 * cy + (c >> 1)) for (cx, cy) in cells for c in range(4)]
    return cells


def dealii_numbering_2d(level, p):
    """index of every lexicographic DoF (row-major [J][I], flattened) in deal.II's numbering"""
    nc = 2**level
    n = nc * p + 1
    num = -np.ones((n, n), dtype=int)
    nxt = 0
    for cx, cy in morton_cells(level):
        i0, j0 = cx * p, cy * p
        for di, dj in [(0, 0), (p, 0), (0, p), (p, p)]:
            if num[j0 + dj, i0 + di] < 0:
                num[j0 + dj, i0 + di] = nxt
                nxt += 1
        for line in range(4):
            pts = {
                0: [(0, j) for j in range(1, p)],
                1: [(p, j) for j in range(1, p)],
                2: [(i, 0) for i in range(1, p)],
                3: [(i, p) for i in range(1, p)],
            }[line]
            if pts and num[j0 + pts[0][1], i0 + pts[0][0]] < 0:
                for i, j in pts:
                    num[j0 + j, i0 + i] = nxt
                    nxt += 1
        for j in range(1, p):
            for i in range(1, p):
                num[j0 + j, i0 + i] = nxt
                nxt += 1
    assert nxt == n * n
    return num.reshape(-1)
