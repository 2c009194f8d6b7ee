#!/usr/bin/env python3
"""Kernel times of BASELINE config 5's shape on one GPU: degree 7, Kershaw mesh, restricted additive Schwarz with
n_overlap = 2 (patches of 10^3 DoFs), Chebyshev degree 3; about 3e7 DoFs (44^3 cells).  Context, not the bench metric."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

pkg = load_package()
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 44
for geometry, weighting, n_overlap in (("kershaw", "ras", 2), ("cartesian", "ras", 2), ("cartesian", "post", 1)):
    c = pkg.Context(degree=7, n_cells=(nc, nc, nc), boundary="dirichlet", geometry=geometry, kershaw_eps=(0.3, 0.3), precond="FDM",
                    n_overlap=n_overlap, weighting=weighting, cheb_degree=3)
    c.set_eigenvalue(1.7)
    b, x, y = c.vector(np.ones(c.n_owned)), c.vector(), c.vector()
    out = {}
    for name, fn in (("A", lambda: c.op_vmult(y, b)), ("P", lambda: c.precond_vmult(y, b)), ("step", lambda: c.cheb_step(x, b))):
        for _ in range(2):
            fn()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        c.synchronize()
        out[name] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"p=7 {geometry} {nc}^3 cells ({c.n_dofs} DoFs) {weighting} n_overlap={n_overlap}: A {out['A']:.3f} ms, P {out['P']:.3f} ms, "
          f"Chebyshev(3) step {out['step']:.3f} ms = {c.n_dofs / out['step'] * 1e3:.3e} DoF/s")
    for v in (b, x, y):
        v.free()
    c.close()
