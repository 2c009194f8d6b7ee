#!/usr/bin/env python3
"""Whole-solve timing (context only, not the bench metric): outer Krylov solver preconditioned by one multigrid
V-cycle with Chebyshev + FDM smoothing, as element_centered_preconditioners_01.cc solve() does for the experiment
files; prints iterations, time to solution (best of N, as solve() reports it) and the V-cycle time.

    python tools/solve_run.py --cells 48 48 48 --geometry cartesian --solver GMRES --degree 1   # amd-opt_fdm1_cart.json
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, nargs=3, default=[48, 48, 48])
    ap.add_argument("--geometry", default="cartesian", choices=["cartesian", "kershaw"])
    ap.add_argument("--solver", default="GMRES", choices=["GMRES", "CG"])
    ap.add_argument("--degree", type=int, default=1, help="Chebyshev degree of the smoother")
    ap.add_argument("--weighting", default="post")
    ap.add_argument("--n-overlap", type=int, default=1)
    ap.add_argument("--patch", default="cell", choices=["cell", "vertex"])
    ap.add_argument("--mg-type", default="ph")
    ap.add_argument("--rel-tol", type=float, default=1e-5)
    ap.add_argument("--tmp-vectors", type=int, default=17)
    ap.add_argument("--best-of", type=int, default=5)
    ap.add_argument("--coarse", default="cg", choices=["cg", "smoother", "direct"])
    ap.add_argument("--keep-basis", action="store_true", help="GMRES keeps the preconditioned basis (asm_solver_params::keep_preconditioned_basis)")
    args = ap.parse_args()
    pkg = load_package()
    prm = pkg.make_params(degree=4, n_cells=args.cells, boundary="dirichlet", geometry=args.geometry, kershaw_eps=(0.3, 0.3),
                          weighting=args.weighting, n_overlap=args.n_overlap, element_centric=args.patch == "cell",
                          cheb_degree=args.degree)
    t0 = time.perf_counter()
    mg = pkg.Multigrid(prm, mg_type=args.mg_type, coarse=args.coarse, coarse_rel_tol=1e-3)
    fine = mg.fine
    fine.synchronize()
    t_setup = time.perf_counter() - t0
    b = fine.vector(np.ones(fine.n_owned))  # f = 1 up to the mass scaling: any smooth right-hand side
    fine.constraints_set_zero(b)
    x = fine.vector()
    best, res = 1e30, None
    for i in range(args.best_of + 1):  # first run = warm-up, as solve() does
        fine.synchronize()
        t0 = time.perf_counter()
        res = fine.solve(x, b, solver=args.solver, preconditioner="Multigrid", mg=mg, rel_tol=args.rel_tol, abs_tol=1e-10,
                         max_n_tmp_vectors=args.tmp_vectors, keep_preconditioned_basis=args.keep_basis)
        fine.synchronize()
        if i > 0:
            best = min(best, time.perf_counter() - t0)
    # one V-cycle
    y = fine.vector()
    for _ in range(2):
        mg.vmult(y, b)
    fine.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mg.vmult(y, b)
    fine.synchronize()
    t_v = (time.perf_counter() - t0) / 5
    levels = [(c.n_dofs) for c in mg.levels]
    print(f"cells {args.cells} {args.geometry} n_dofs {fine.n_dofs} levels {levels} {args.solver} k={args.degree} {args.weighting} "
          f"n_overlap={args.n_overlap} patch={args.patch} coarse={args.coarse}{' keep-basis' if args.keep_basis else ''}: it {res.n_iterations} converged {res.converged} "
          f"time {best:.4e} s  (V-cycle {t_v * 1e3:.3f} ms, setup {t_setup:.2f} s)")


if __name__ == "__main__":
    main()
