"""GPU parity: HIP path (through the C ABI) against the numpy oracle on the same
seeded inputs.  Tolerance: relative l2 difference < 1e-12 (BASELINE.json north_star:
"residual match to CPU reference <1e-12 relative"); the generic kernels add with
fp64 atomics, so results are reproducible only to rounding."""
import numpy as np
import pytest

import asm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-12
BCMAP = {"dirichlet": orc.DIRICHLET, "periodic": orc.PERIODIC, "neumann": orc.NEUMANN}


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make_pair(pkg, nc, p, bc, geometry="cartesian", eps=(1.0, 1.0), extent=(1.0, 1.0, 1.0), **kw):
    """(oracle mesh, GPU context) for the same problem"""
    if isinstance(bc, str):
        bc = [bc] * 3
    mapfn = None
    if geometry == "kershaw":
        mapfn = lambda pts: orc.kershaw(eps[0], eps[1], pts)  # noqa: E731
    mesh = orc.Mesh(nc, p, [BCMAP[b] for b in bc], h=[extent[d] / nc[d] for d in range(3)], mapfn=mapfn)
    ctx = pkg.Context(
        degree=p, n_cells=nc, boundary=list(bc), geometry=geometry, kershaw_eps=eps, extent=extent, **kw
    )
    return mesh, ctx


@pytest.mark.parametrize(
    "nc,p,bc,geometry",
    [
        ((3, 2, 2), 4, "periodic", "cartesian"),
        ((3, 3, 2), 4, "dirichlet", "cartesian"),
        ((4, 3, 2), 2, ["dirichlet", "periodic", "neumann"], "cartesian"),
        ((6, 2, 2), 3, "dirichlet", "kershaw"),
        ((5, 4, 3), 1, "dirichlet", "cartesian"),
        ((2, 2, 2), 7, "dirichlet", "cartesian"),
    ],
)
def test_op_vmult(pkg, nc, p, bc, geometry):
    mesh, ctx = make_pair(pkg, nc, p, bc, geometry, eps=(0.3, 0.3), extent=(1.0, 1.0, 1.0), kernel="generic")
    with ctx:
        assert ctx.n_dofs == mesh.n_dofs
        rng = np.random.default_rng(2024)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        ref = orc.LaplaceOperator(mesh).vmult(src)
        x, y = ctx.vector(src), ctx.vector()
        ctx.op_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("weighting", ["none", "pre", "post", "symm", "ras"])
@pytest.mark.parametrize(
    "nc,p,bc,geometry,n_overlap,element_centric",
    [
        ((3, 2, 2), 4, "periodic", "cartesian", 1, True),
        ((3, 3, 2), 4, "dirichlet", "cartesian", 1, True),
        ((3, 3, 3), 3, "dirichlet", "cartesian", 2, True),
        ((3, 3, 3), 3, "periodic", "cartesian", 3, True),
        ((3, 3, 2), 3, "dirichlet", "cartesian", 1, False),
        ((6, 2, 2), 4, "dirichlet", "kershaw", 1, True),
        ((6, 2, 2), 2, "dirichlet", "kershaw", 2, True),
        ((6, 2, 2), 3, "dirichlet", "kershaw", 1, False),
    ],
)
def test_precond_vmult(pkg, nc, p, bc, geometry, n_overlap, element_centric, weighting):
    mesh, ctx = make_pair(
        pkg,
        nc,
        p,
        bc,
        geometry,
        eps=(0.3, 0.3),
        kernel="generic",
        n_overlap=n_overlap,
        weighting=weighting,
        element_centric=element_centric,
    )
    with ctx:
        rng = np.random.default_rng(7)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        P = orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric)
        ref = P.vmult(src)
        x, y = ctx.vector(src), ctx.vector()
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize(
    "nc,p,bc,weighting,precond",
    [
        ((3, 2, 2), 4, "periodic", "post", "FDM"),
        ((3, 3, 2), 4, "dirichlet", "post", "FDM"),
        ((3, 3, 2), 4, "dirichlet", "symm", "FDM"),
        ((4, 4, 4), 1, "dirichlet", "post", "Diagonal"),
    ],
)
def test_chebyshev(pkg, nc, p, bc, weighting, precond, degree):
    mesh, ctx = make_pair(pkg, nc, p, bc, kernel="generic", weighting=weighting, precond=precond, cheb_degree=degree)
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, 1, weighting) if precond == "FDM" else orc.JacobiPreconditioner(op)
        cheb = orc.Chebyshev(op, prec, degree=degree)
        lam = ctx.estimate_eigenvalues()
        assert abs(lam - cheb.lambda_hat) < 1e-9 * cheb.lambda_hat
        ctx.set_eigenvalue(cheb.lambda_hat)
        rng = np.random.default_rng(11)
        b = rng.uniform(-1, 1, mesh.n_dofs)
        x0 = rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


# ---------------------------------------------------------------------------
# tiled owner-computes kernels (degree 4, n_overlap 1): same oracle, same tolerance
# ---------------------------------------------------------------------------
TILED_CASES = [
    ((3, 2, 2), "periodic", "cartesian"),
    ((3, 3, 2), "dirichlet", "cartesian"),
    ((17, 4, 3), ["dirichlet", "periodic", "neumann"], "cartesian"),
    ((4, 16, 5), ["periodic", "dirichlet", "dirichlet"], "cartesian"),
    ((6, 2, 2), "dirichlet", "kershaw"),
    ((18, 16, 2), "dirichlet", "kershaw"),
    # cell counts of 1 modulo the owned cells of a tile (11 in x, 7 in y for 8-row tiles): at a physical boundary tile 0
    # starts at cell 0 and owns its first column / row as well (Tiling::x_first, y_first) - one tile owning everything,
    # two tiles per direction, a last tile with a single column
    ((12, 8, 2), "dirichlet", "cartesian"),
    ((23, 15, 2), "dirichlet", "cartesian"),
    ((13, 9, 3), ["neumann", "dirichlet", "periodic"], "cartesian"),
    ((12, 22, 2), "dirichlet", "kershaw"),
]


@pytest.mark.parametrize("weighting", ["none", "pre", "post", "symm", "ras"])
@pytest.mark.parametrize("nc,bc,geometry", TILED_CASES)
def test_precond_vmult_tiled(pkg, nc, bc, geometry, weighting):
    mesh, ctx = make_pair(pkg, nc, 4, bc, geometry, eps=(0.3, 0.3), kernel="tiled", weighting=weighting)
    with ctx:
        assert ctx.kernel_variant() == "tiled"
        rng = np.random.default_rng(7)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        ref = orc.ASPoissonPreconditioner(mesh, 1, weighting).vmult(src)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 123.0))  # dst needs no zeroing
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
@pytest.mark.parametrize("nc,bc,geometry", TILED_CASES[:5] + TILED_CASES[6:])
def test_chebyshev_tiled(pkg, nc, bc, geometry, degree):
    mesh, ctx = make_pair(pkg, nc, 4, bc, geometry, eps=(0.3, 0.3), kernel="tiled", weighting="post", cheb_degree=degree)
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, 1, "post")
        cheb = orc.Chebyshev(op, prec, degree=degree, lambda_max=1.9)
        ctx.set_eigenvalue(1.9)
        rng = np.random.default_rng(11)
        b = rng.uniform(-1, 1, mesh.n_dofs)
        x0 = rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


@pytest.mark.parametrize("rows", ["0", "9", "10", "11"])
@pytest.mark.parametrize("nc,bc", [((13, 9, 3), ["neumann", "dirichlet", "periodic"]), ((12, 10, 2), "dirichlet"), ((23, 21, 2), "dirichlet"),
                                   ((5, 19, 4), "periodic")])
def test_laplace_row_variants(pkg, monkeypatch, nc, bc, rows):
    """every variant of the tiled Laplace sweep (8 rows per workgroup = first generation, 9-11 = second generation;
    ASM_TILED_ROWS fixes what the context otherwise measures), on meshes whose row counts are 1 modulo the owned rows of
    the variants (8, 9, 10 owned rows: tile 0 at a physical boundary owns its first row too) and on a periodic one"""
    monkeypatch.setenv("ASM_TILED_ROWS", rows)
    mesh, ctx = make_pair(pkg, nc, 4, bc, kernel="tiled", weighting="post", cheb_degree=2)
    with ctx:
        assert ctx.tiled_laplace_rows() == (8 if rows == "0" else int(rows))
        op = orc.LaplaceOperator(mesh)
        rng = np.random.default_rng(3)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 5.0))
        ctx.op_vmult(y, x)
        assert rel(y.get(), op.vmult(src)) < TOL
        cheb = orc.Chebyshev(op, orc.ASPoissonPreconditioner(mesh, 1, "post"), degree=2, lambda_max=1.9)
        ctx.set_eigenvalue(1.9)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL
