"""Races in the LDS protocols of the tiled kernels (single transpose buffer, y-merge slots, staged image
reuse) would show as run-to-run differences: repeat a Chebyshev step on the full-size problems and
compare the results bitwise."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
for boundary, cells, degree in (("dirichlet", (80, 80, 80), 1), ("periodic", (128, 64, 64), 3), ("dirichlet", (47, 33, 21), 2)):
    ctx = pkg.Context(degree=4, n_cells=cells, boundary=boundary, cheb_degree=degree)
    ctx.set_eigenvalue(1.7)
    rng = np.random.default_rng(1)
    b = ctx.vector(rng.standard_normal(ctx.n_owned))
    x0 = rng.standard_normal(ctx.n_owned)
    ref = None
    for rep in range(8):
        x = ctx.vector(x0)
        for _ in range(3):
            ctx.cheb_step(x, b)
        out = x.get()
        if ref is None:
            ref = out
        elif not np.array_equal(ref, out):
            print("MISMATCH", boundary, cells, degree, "rep", rep, "max diff", np.abs(ref - out).max())
            sys.exit(1)
        x.free()
    print("bitwise reproducible:", boundary, cells, "degree", degree, "norm", float(np.linalg.norm(ref)))
    ctx.close()

# the multi-rank path on one GPU (ASM_SELF_INTERFACE: the rank's ghost exchange with itself through RCCL, overlapped with
# the sweeps, ghosts kept between steps): a missing wait between the streams would show the same way
os.environ["ASM_SELF_INTERFACE"] = "1"
ctx = pkg.Context(degree=4, n_cells=(66, 44, 40), boundary="periodic", cheb_degree=3)
ctx.comm_init(pkg.comm_unique_id())
ctx.set_keep_ghosts(True)
ctx.set_eigenvalue(1.7)
rng = np.random.default_rng(1)
b = ctx.vector(rng.standard_normal(ctx.n_owned))
x0 = rng.standard_normal(ctx.n_owned)
ref = None
for rep in range(8):
    x = ctx.vector(x0)
    for _ in range(3):
        ctx.cheb_step(x, b)
    out = x.get()
    if ref is None:
        ref = out
    elif not np.array_equal(ref, out):
        print("MISMATCH in the self-interface (RCCL) path, rep", rep, "max diff", np.abs(ref - out).max())
        sys.exit(1)
    x.free()
print("bitwise reproducible: self-interface (RCCL) path, periodic (66, 44, 40) degree 3, keep_ghosts, norm", float(np.linalg.norm(ref)))
ctx.close()
