"""GPU parity of the BASELINE.json configurations that need the 7^3 / 10^3 patch kernels and the
deformed-cell operator, against the numpy oracle (same tolerance as test_gpu_parity.py):

  config 3  Kershaw, p=4, n_overlap=2 (m=7), post weights, Chebyshev k=2   (experiments/large-scaling-opt_fdm2.json:13-25)
  config 4  p=4, vertex-star patches (m=7), symm weights, Chebyshev k=2    (experiments/large-scaling-opt_fdmv.json:13-25)
  config 5  Kershaw, p=7, RAS, n_overlap=2 (m=10), Chebyshev k=3           (experiments/large-scaling-opt_ras.json:13-25)

and the bench workload at full size (80^3 cells) against the threaded C++ restatement.
Mesh sizes are chosen so that the generic FDM kernel meets all its matrix paths: blocks of cells
that share a parity-sorted instance (even-odd sweeps), share a boundary instance (scalar operands),
or mix instances (LDS matrices)."""
import os

import numpy as np
import pytest

import asm_oracle as orc
from test_gpu_parity import TOL, make_pair, rel

pytestmark = pytest.mark.gpu

# (nc, bc, geometry, n_overlap, element_centric)
M7_CASES = [
    ((5, 4, 3), "dirichlet", "cartesian", 2, True),
    ((4, 3, 3), "periodic", "cartesian", 2, True),
    ((5, 3, 3), ["periodic", "dirichlet", "neumann"], "cartesian", 2, True),
    ((6, 2, 2), "dirichlet", "kershaw", 2, True),
    ((5, 4, 3), "dirichlet", "cartesian", 1, False),
    ((4, 3, 3), "periodic", "cartesian", 1, False),
    ((6, 3, 2), "dirichlet", "kershaw", 1, False),
    # deformed meshes with cells all of whose patch DoFs exist: the per-cell-matrix slice-per-lane kernel takes those
    ((12, 3, 3), "dirichlet", "kershaw", 2, True),
    ((4, 4, 3), ["dirichlet", "neumann", "dirichlet"], "kershaw", 1, False),
]


@pytest.fixture(params=[False, True], ids=["store_gather", "atomics"])
def fdm_atomics(request):
    """the generic FDM sweep has two scatter forms (context.hip: patch_tmp / ASM_FDM_ATOMICS)"""
    old = os.environ.get("ASM_FDM_ATOMICS")
    if request.param:
        os.environ["ASM_FDM_ATOMICS"] = "1"
    else:
        os.environ.pop("ASM_FDM_ATOMICS", None)
    yield request.param
    if old is None:
        os.environ.pop("ASM_FDM_ATOMICS", None)
    else:
        os.environ["ASM_FDM_ATOMICS"] = old


@pytest.mark.parametrize("weighting", ["none", "pre", "post", "symm", "ras"])
@pytest.mark.parametrize("nc,bc,geometry,n_overlap,element_centric", M7_CASES)
def test_precond_vmult_p4_m7(pkg, fdm_atomics, nc, bc, geometry, n_overlap, element_centric, weighting):
    mesh, ctx = make_pair(pkg, nc, 4, bc, geometry, eps=(0.3, 0.3), n_overlap=n_overlap, weighting=weighting,
                          element_centric=element_centric)
    with ctx:
        src = np.random.default_rng(7).uniform(-1, 1, mesh.n_dofs)
        ref = orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric).vmult(src)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 123.0))
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize(
    "nc,bc,geometry,n_overlap,element_centric,weighting",
    [
        ((6, 2, 2), "dirichlet", "kershaw", 2, True, "post"),  # config 3 (file variant)
        ((5, 4, 3), "dirichlet", "cartesian", 2, True, "post"),
        ((4, 3, 3), "periodic", "cartesian", 1, False, "symm"),  # config 4
        ((6, 3, 2), "dirichlet", "kershaw", 1, False, "symm"),  # the file behind config 4 is a Kershaw mesh
        ((6, 2, 2), "dirichlet", "kershaw", 2, True, "ras"),
    ],
)
@pytest.mark.parametrize("degree", [1, 2, 3])
def test_chebyshev_p4_m7(pkg, nc, bc, geometry, n_overlap, element_centric, weighting, degree):
    mesh, ctx = make_pair(pkg, nc, 4, bc, geometry, eps=(0.3, 0.3), n_overlap=n_overlap, weighting=weighting,
                          element_centric=element_centric, cheb_degree=degree)
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric)
        cheb = orc.Chebyshev(op, prec, degree=degree, lambda_max=1.7)
        ctx.set_eigenvalue(1.7)
        rng = np.random.default_rng(11)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


@pytest.mark.parametrize("lz", [None, "3"])
@pytest.mark.parametrize(
    "nc,bc,geometry,n_overlap,element_centric,weighting",
    [
        ((20, 9, 13), "dirichlet", "cartesian", 2, True, "post"),    # box 18 x 7 x 11: columns (9, 9) x (4, 3), z chunks (6, 5)
        ((20, 10, 14), "periodic", "cartesian", 1, False, "symm"),   # vertex star, every cell in the box, wrap-around strips
        ((19, 9, 13), ["periodic", "dirichlet", "neumann"], "cartesian", 2, True, "none"),
        ((20, 6, 13), "dirichlet", "kershaw", 2, True, "post"),      # per-cell eigen-decompositions
        ((11, 5, 8), "dirichlet", "cartesian", 2, True, "pre"),      # one column, a ragged second chunk
    ],
)
def test_m7_accumulate_multi_column(pkg, monkeypatch, nc, bc, geometry, n_overlap, element_centric, weighting, lz):
    """the accumulate form of the 7^3 sweep (fdm_slice7_acc_kernel) on meshes with several columns and z chunks, so that the
    strips shared between workgroups (global atomics on values the init kernel wrote) and the DoFs a workgroup stores alone
    both occur in every direction; plain product and the three fused Chebyshev updates (degree-3 step: first sweep,
    x + f2 P^-1 r, three-term update onto x_old); ASM_S7_LZ=3 makes every chunk three layers"""
    if lz:
        monkeypatch.setenv("ASM_S7_LZ", lz)
    mesh, ctx = make_pair(pkg, nc, 4, bc, geometry, eps=(0.3, 0.3), n_overlap=n_overlap, weighting=weighting,
                          element_centric=element_centric, cheb_degree=3)
    with ctx:
        rng = np.random.default_rng(5)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        prec = orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 123.0))
        ctx.precond_vmult(y, x)
        assert rel(y.get(), prec.vmult(src)) < TOL
        cheb = orc.Chebyshev(orc.LaplaceOperator(mesh), prec, degree=3, lambda_max=1.7)
        ctx.set_eigenvalue(1.7)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


@pytest.mark.parametrize("nc,bc,geometry", [((3, 2, 2), "dirichlet", "kershaw"), ((2, 3, 2), ["dirichlet", "neumann", "dirichlet"], "kershaw"),
                                            ((4, 4, 4), "dirichlet", "kershaw")])
def test_op_vmult_p7_deformed(pkg, nc, bc, geometry):
    mesh, ctx = make_pair(pkg, nc, 7, bc, geometry, eps=(0.3, 0.3))
    with ctx:
        src = np.random.default_rng(2024).uniform(-1, 1, mesh.n_dofs)
        ref = orc.LaplaceOperator(mesh).vmult(src)
        x, y = ctx.vector(src), ctx.vector()
        ctx.op_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("weighting", ["ras", "post", "symm"])
@pytest.mark.parametrize("nc,bc,geometry", [((3, 2, 2), "dirichlet", "kershaw"), ((3, 3, 2), "dirichlet", "cartesian"),
                                            ((2, 2, 2), "periodic", "cartesian")])
def test_precond_vmult_p7_m10(pkg, fdm_atomics, nc, bc, geometry, weighting):
    """config 5: p=7, n_overlap=2 -> 10^3 patches"""
    mesh, ctx = make_pair(pkg, nc, 7, bc, geometry, eps=(0.3, 0.3), n_overlap=2, weighting=weighting)
    with ctx:
        src = np.random.default_rng(7).uniform(-1, 1, mesh.n_dofs)
        ref = orc.ASPoissonPreconditioner(mesh, 2, weighting).vmult(src)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 123.0))
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


def test_precond_vmult_p7_vertex_star_m13(pkg):
    mesh, ctx = make_pair(pkg, (3, 2, 2), 7, "dirichlet", "kershaw", eps=(0.3, 0.3), weighting="symm", element_centric=False)
    with ctx:
        src = np.random.default_rng(7).uniform(-1, 1, mesh.n_dofs)
        ref = orc.ASPoissonPreconditioner(mesh, 1, "symm", False).vmult(src)
        x, y = ctx.vector(src), ctx.vector()
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


def test_chebyshev_p7_ras_config5(pkg):
    nc = (3, 2, 2)
    mesh, ctx = make_pair(pkg, nc, 7, "dirichlet", "kershaw", eps=(0.3, 0.3), n_overlap=2, weighting="ras", cheb_degree=3)
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, 2, "ras")
        cheb = orc.Chebyshev(op, prec, degree=3, lambda_max=1.5)
        ctx.set_eigenvalue(1.5)
        rng = np.random.default_rng(11)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


@pytest.mark.parametrize("p,nc,geometry,n_overlap", [(4, (6, 4, 4), "cartesian", 2), (4, (6, 2, 2), "kershaw", 2), (3, (4, 4, 4), "cartesian", 1), (2, (6, 6, 4), "cartesian", 2)])
def test_ras_direct_store_equals_atomics(pkg, monkeypatch, p, nc, geometry, n_overlap):
    """restricted additive Schwarz on one rank: the patch kernel that stores the updated values itself (every DoF has
    exactly one contributing patch) against the zero + atomics + update passes form (ASM_RAS_ATOMICS), both against the oracle"""
    out = []
    rng = np.random.default_rng(13)
    for atomics in (False, True):
        if atomics:
            monkeypatch.setenv("ASM_RAS_ATOMICS", "1")
        mesh, ctx = make_pair(pkg, nc, p, ["dirichlet", "neumann", "dirichlet"], geometry, eps=(0.3, 0.3), n_overlap=n_overlap,
                              weighting="ras", cheb_degree=3)
        with ctx:
            if not out:
                b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
                op = orc.LaplaceOperator(mesh)
                cheb = orc.Chebyshev(op, orc.ASPoissonPreconditioner(mesh, n_overlap, "ras"), degree=3, lambda_max=1.5)
                ref = (cheb.vmult(b), cheb.step(x0, b))
            ctx.set_eigenvalue(1.5)
            bv, xv, yv = ctx.vector(b), ctx.vector(), ctx.vector()
            ctx.precond_vmult(yv, bv)
            ctx.cheb_vmult(xv, bv)
            r1 = xv.get()
            xv.set(x0)
            ctx.cheb_step(xv, bv)
            out.append((yv.get(), r1, xv.get()))
    for k in range(3):
        assert rel(out[0][k], out[1][k]) < TOL
    assert rel(out[0][1], ref[0]) < TOL and rel(out[0][2], ref[1]) < TOL


# ---------------------------------------------------------------------------
# the bench workload at full size against the threaded C++ restatement (itself checked against the
# numpy oracle in tests/test_oracle_cpu_port.py)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nc,periodic,degree", [((80, 80, 80), False, 1), ((80, 80, 80), False, 3), ((64, 32, 32), True, 2)])
def test_full_size_bench_workload(pkg, nc, periodic, degree):
    import subprocess

    subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")],
                   check=True, capture_output=True)
    import asm_oracle_cpu

    lam = 1.7
    sm = asm_oracle_cpu.CpuSmoother(nc, periodic, degree=degree, lambda_max=lam, n_threads=min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(5)
    b, x0 = rng.uniform(-1, 1, sm.n_dofs), rng.uniform(-1, 1, sm.n_dofs)
    ref = sm.step(x0, b)
    with pkg.Context(degree=4, n_cells=nc, boundary="periodic" if periodic else "dirichlet", cheb_degree=degree,
                     kernel="tiled") as ctx:
        assert ctx.n_dofs == sm.n_dofs and ctx.kernel_variant() == "tiled"
        ctx.set_eigenvalue(lam)
        bv, xv = ctx.vector(b), ctx.vector(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), ref) < TOL


@pytest.mark.parametrize("degree", [1, 2])
def test_full_size_kershaw_workload(pkg, degree):
    """BASELINE config 3 as worded (Kershaw eps = 0.3, p = 4, ~3e7 DoFs, FDM n_overlap = 1, post weights) at full size:
    78^3 cells = 30 664 297 DoFs, one Chebyshev step against the threaded C++ restatement's deformed-mesh leg (checked
    against the numpy oracle in tests/test_oracle_cpu_port.py; its per-cell tables come from the numpy oracle's array
    code, the eigen-decompositions batched: tests/test_oracle_cpu_port.py::test_batched_eigen_decompositions)"""
    import subprocess

    subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")],
                   check=True, capture_output=True)
    import asm_oracle_cpu

    nc, lam = (78, 78, 78), 1.8
    mesh = orc.Mesh(nc, 4, [orc.DIRICHLET] * 3, mapfn=lambda pts: orc.kershaw(0.3, 0.3, pts))
    sm = asm_oracle_cpu.CpuSmootherDeformed(mesh, degree=degree, lambda_max=lam, n_threads=min(32, os.cpu_count() or 1))
    rng = np.random.default_rng(7)
    b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
    ref = sm.step(x0, b)
    del sm, mesh
    with pkg.Context(degree=4, n_cells=nc, boundary="dirichlet", geometry="kershaw", kershaw_eps=(0.3, 0.3), cheb_degree=degree) as ctx:
        assert ctx.n_dofs == len(b)
        ctx.set_eigenvalue(lam)
        bv, xv = ctx.vector(b), ctx.vector(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), ref) < TOL


# ---------------------------------------------------------------------------
# "polynomial type" = "4th kind" (precondition.templates.h:135-152) and the fused vmult(dst, src, pre, post) surface
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("degree", [1, 2, 3, 5])
@pytest.mark.parametrize(
    "nc,p,bc,geometry,kernel,precond,weighting",
    [
        ((13, 5, 3), 4, "dirichlet", "cartesian", "tiled", "FDM", "post"),
        ((3, 2, 2), 4, "periodic", "cartesian", "tiled", "FDM", "symm"),
        ((6, 2, 2), 3, "dirichlet", "kershaw", "generic", "FDM", "post"),
        ((4, 4, 4), 1, "dirichlet", "cartesian", "generic", "Diagonal", "post"),
    ],
)
def test_chebyshev_fourth_kind(pkg, nc, p, bc, geometry, kernel, precond, weighting, degree):
    mesh, ctx = make_pair(pkg, nc, p, bc, geometry, eps=(0.3, 0.3), kernel=kernel, weighting=weighting, precond=precond,
                          cheb_degree=degree, options={"polynomial type": "4th kind"})
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, 1, weighting) if precond == "FDM" else orc.JacobiPreconditioner(op)
        cheb = orc.Chebyshev(op, prec, degree=degree, lambda_max=1.6, polynomial_type="4th kind")
        ctx.set_eigenvalue(1.6)
        rng = np.random.default_rng(11)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


@pytest.mark.parametrize(
    "nc,p,bc,geometry,kernel,n_overlap",
    [((13, 5, 3), 4, "dirichlet", "cartesian", "tiled", 1), ((6, 2, 2), 4, "dirichlet", "kershaw", "auto", 1),
     ((5, 4, 3), 4, "dirichlet", "cartesian", "auto", 2), ((3, 3, 3), 2, "periodic", "cartesian", "generic", 1)],
)
def test_fused_overloads_reproduce_chebyshev(pkg, nc, p, bc, geometry, kernel, n_overlap):
    """a damped relaxation and the three-term recurrence written by the caller over asm_op_vmult_fused /
    asm_precond_vmult_fused against the oracle's Chebyshev iteration"""
    mesh, ctx = make_pair(pkg, nc, p, bc, geometry, eps=(0.3, 0.3), kernel=kernel, n_overlap=n_overlap, cheb_degree=3)
    with ctx:
        op = orc.LaplaceOperator(mesh)
        prec = orc.ASPoissonPreconditioner(mesh, n_overlap, "post")
        cheb = orc.Chebyshev(op, prec, degree=3, lambda_max=1.8)
        ctx.set_eigenvalue(1.8)
        par = ctx.cheb_parameters()
        theta, delta = par["theta"], par["delta"]
        rng = np.random.default_rng(3)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, x, x_old, r = ctx.vector(b), ctx.vector(x0), ctx.vector(), ctx.vector()
        ctx.op_vmult_fused("residual", r, x, b=bv)
        ctx.precond_vmult_fused("relax", x_old, r, x=x, f1=1.0, f2=1.0 / theta)
        x, x_old = x_old, x
        rhok, sigma = delta / theta, theta / delta
        for _ in range(2):
            rhokp = 1.0 / (2.0 * sigma - rhok)
            f1, f2, rhok = rhokp * rhok, 2.0 * rhokp / delta, rhokp
            ctx.op_vmult_fused("residual", r, x, b=bv)
            ctx.precond_vmult_fused("cheb_next", x_old, r, x=x, x_old=x_old, f1=f1, f2=f2)
            x, x_old = x_old, x
        assert rel(x.get(), cheb.step(x0, b)) < TOL
        # plain forms and argument checks
        y = ctx.vector()
        ctx.op_vmult_fused("none", y, x)
        assert rel(y.get(), op.vmult(x.get())) < TOL
        ctx.precond_vmult_fused("cheb_first", y, bv, f2=0.5)
        assert rel(y.get(), 0.5 * prec.vmult(b, unit_constrained=True)) < TOL
        with pytest.raises(RuntimeError):
            ctx.op_vmult_fused("residual", y, x)  # b missing
        with pytest.raises(RuntimeError):
            ctx.precond_vmult_fused("cheb_next", x, r, x=x, x_old=x_old, f1=0.1, f2=0.1)  # dst aliases x
        assert ctx.is_symmetric() is False
