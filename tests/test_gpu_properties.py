"""Size-independent properties at the benchmark's full size (80^3 cells, p = 4: 3.3e7 DoFs), where
the oracle is too slow to serve as a checker, and tile / chunk edge cases of the tiled kernels at
sizes the oracle still handles."""
import numpy as np
import pytest

import asm_oracle as orc
from test_gpu_parity import BCMAP, make_pair, rel

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def pkg():
    from conftest import load_package

    return load_package()


def test_full_size_symmetry_and_linearity(pkg):
    ctx = pkg.Context(degree=4, n_cells=(80, 80, 80), boundary="dirichlet", precond="FDM", n_overlap=1, weighting="symm", cheb_degree=2)
    with ctx:
        assert ctx.kernel_variant() == "tiled"
        n = ctx.n_owned
        rng = np.random.default_rng(5)
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        xv, yv, t = ctx.vector(x), ctx.vector(y), ctx.vector()
        ctx.op_vmult(t, xv)
        ax = t.get()
        ctx.op_vmult(t, yv)
        ay = t.get()
        # A is symmetric (constrained rows and columns are zero in the plain product)
        assert abs(ax @ y - x @ ay) < TOL * np.linalg.norm(ax) * np.linalg.norm(y)
        # and positive on vectors that vanish on the boundary: x^T A x > 0
        assert ax @ x > 0
        ctx.precond_vmult(t, xv)
        px = t.get()
        ctx.precond_vmult(t, yv)
        py = t.get()
        # symmetric weighting gives a symmetric preconditioner (matrix_free.h:925-986)
        assert abs(px @ y - x @ py) < TOL * np.linalg.norm(px) * np.linalg.norm(y)
        # the Chebyshev step is affine in (x0, b): S(x0 + x1, b0 + b1) = S(x0, b0) + S(x1, b1) - S(0, 0), S(0, 0) = 0
        ctx.set_eigenvalue(1.8)
        zv = ctx.vector(np.zeros(n))
        a = ctx.vector(x)
        ctx.cheb_step(a, yv)  # S(x, y)
        sxy = a.get()
        a.set(x)
        ctx.cheb_step(a, zv)  # S(x, 0)
        sx0 = a.get()
        a.set(np.zeros(n))
        ctx.cheb_step(a, yv)  # S(0, y)
        s0y = a.get()
        assert rel(sx0 + s0y, sxy) < TOL


def test_full_size_periodic_constant_nullspace(pkg):
    ctx = pkg.Context(degree=4, n_cells=(80, 80, 80), boundary="periodic", precond="FDM", n_overlap=1, weighting="post")
    with ctx:
        n = ctx.n_owned
        one, t = ctx.vector(np.ones(n)), ctx.vector()
        ctx.op_vmult(t, one)
        # rows of the periodic Laplacian sum to zero; |A|_inf is O(h) for the 3-D stiffness matrix
        assert np.abs(t.get()).max() < 1e-13
        rng = np.random.default_rng(9)
        x = rng.uniform(-1, 1, n)
        ctx.op_vmult(t, ctx.vector(x))
        assert abs(t.get().sum()) < 1e-12 * np.abs(t.get()).sum()


@pytest.mark.parametrize(
    "nc,bc,extent",
    [
        ((25, 17, 9), "dirichlet", (1.0, 1.0, 1.0)),  # 3 x 3 tiles, one cell layer per z chunk
        ((12, 8, 11), ["periodic", "neumann", "dirichlet"], (2.0, 1.0, 0.5)),  # exactly one tile + halo wrap, anisotropic cells
        ((23, 15, 4), "periodic", (1.0, 0.7, 0.3)),  # tile edges on the periodic seam
    ],
)
def test_tile_and_chunk_edges(pkg, nc, bc, extent):
    mesh, ctx = make_pair(pkg, nc, 4, bc, "cartesian", extent=extent, kernel="tiled", weighting="post", cheb_degree=2)
    with ctx:
        rng = np.random.default_rng(13)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, -7.0))
        op = orc.LaplaceOperator(mesh)
        ctx.op_vmult(y, x)
        assert rel(y.get(), op.vmult(src)) < TOL
        prec = orc.ASPoissonPreconditioner(mesh, 1, "post")
        ctx.precond_vmult(y, x)
        assert rel(y.get(), prec.vmult(src)) < TOL
        cheb = orc.Chebyshev(op, prec, degree=2, lambda_max=1.9)
        ctx.set_eigenvalue(1.9)
        x0 = rng.uniform(-1, 1, mesh.n_dofs)
        xv = ctx.vector(x0)
        ctx.cheb_step(xv, x)
        assert rel(xv.get(), cheb.step(x0, src)) < TOL


@pytest.mark.parametrize(
    "nc,bc",
    [
        ((1, 1, 1), "dirichlet"),
        ((2, 1, 1), "dirichlet"),
        ((1, 1, 3), ["neumann", "dirichlet", "dirichlet"]),
        ((2, 2, 2), "periodic"),
        ((12, 1, 1), "dirichlet"),
        ((1, 8, 1), "neumann"),
    ],
)
def test_tiny_meshes(pkg, nc, bc):
    """fewer cells than a tile, a wave or a z chunk holds"""
    mesh, ctx = make_pair(pkg, nc, 4, bc, "cartesian", kernel="tiled", weighting="post", cheb_degree=2)
    with ctx:
        rng = np.random.default_rng(3)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        x, y = ctx.vector(src), ctx.vector()
        op, prec = orc.LaplaceOperator(mesh), orc.ASPoissonPreconditioner(mesh, 1, "post")
        ctx.op_vmult(y, x)
        assert rel(y.get(), op.vmult(src)) < TOL
        ctx.precond_vmult(y, x)
        assert rel(y.get(), prec.vmult(src)) < TOL
        ctx.set_eigenvalue(1.9)
        cheb = orc.Chebyshev(op, prec, degree=2, lambda_max=1.9)
        x0 = rng.uniform(-1, 1, mesh.n_dofs)
        xv = ctx.vector(x0)
        ctx.cheb_step(xv, x)
        assert rel(xv.get(), cheb.step(x0, src)) < TOL
