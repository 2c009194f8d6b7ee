"""GPU parity of the smoother's callers (transfers, V-cycle, coarse solver, outer Krylov solvers)
against the oracle twin oracle/asm_oracle_mg.py, which tests/test_oracle_mg_golden.py pins on the reference's own
iteration-count fixtures.  Everything goes through the C ABI (asm_mg_*, asm_solve)."""
import numpy as np
import pytest

import asm_oracle as orc
import asm_oracle_mg as omg
from test_gpu_parity import BCMAP, TOL, rel

pytestmark = pytest.mark.gpu


def fine_params(pkg, nc, p, bc, geometry="cartesian", **kw):
    return pkg.make_params(degree=p, n_cells=nc, boundary=[bc] * 3 if isinstance(bc, str) else list(bc), geometry=geometry,
                           kershaw_eps=(0.3, 0.3), **kw)


def oracle_hierarchy(nc, p, bc, mg_type, weighting, n_overlap, degree, lams, geometry="cartesian", coarse="smoother",
                     coarse_rel_tol=1e-3, one_sided=False, p_sequence="bisect", polynomial_type="1st kind", element_centric=True,
                     intermediate=None, n_coarse_cycles=1, keep_duplicate=True):
    """the oracle twin of asm_mg_create: "mg smoother" = Chebyshev(degree) over FDM on the levels above the intermediate
    level, `intermediate` = (inner "FDM" | "Diagonal", Chebyshev degree) or None (= the mg smoother) on the levels up to
    it; vertex-star patches become the diagonal smoother at degree 1 (as the library does)"""
    bcs = [BCMAP[b] for b in ([bc] * 3 if isinstance(bc, str) else bc)]
    mapfn = (lambda pts: orc.kershaw(0.3, 0.3, pts)) if geometry == "kershaw" else None

    def cheb(level, op, prec, deg):
        return orc.Chebyshev(op, prec, degree=deg, lambda_max=None if lams is None else lams[level], polynomial_type=polynomial_type)

    def smoother(level, mesh, op):
        if not element_centric and mesh.p < 2:
            return cheb(level, op, orc.JacobiPreconditioner(op), degree)
        return cheb(level, op, orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric), degree)

    def inter(level, mesh, op):
        if intermediate is None:
            return smoother(level, mesh, op)
        if intermediate[0] == "Diagonal":
            return cheb(level, op, orc.JacobiPreconditioner(op), intermediate[1])
        if not element_centric and mesh.p < 2:
            return cheb(level, op, orc.JacobiPreconditioner(op), intermediate[1])
        return cheb(level, op, orc.ASPoissonPreconditioner(mesh, n_overlap, weighting, element_centric), intermediate[1])

    def coarse_factory(mesh, op):
        if coarse == "smoother":
            return smoother(0, mesh, op)
        if coarse == "direct":
            return omg.CoarseDirect(op)
        return omg.CoarseCG(op, rel_tol=coarse_rel_tol)

    return omg.build_multigrid(nc, p, bcs, smoother, coarse_factory, mg_type=mg_type, p_sequence=p_sequence, mapfn=mapfn,
                               one_sided=one_sided, intermediate_factory=inter, n_coarse_cycles=n_coarse_cycles,
                               keep_duplicate=keep_duplicate)


@pytest.mark.parametrize(
    "nc,p,bc,mg_type,p_sequence",
    [
        ((4, 2, 2), 4, "dirichlet", "p", "bisect"),
        ((4, 2, 2), 7, ["dirichlet", "neumann", "dirichlet"], "p", "bisect"),
        ((2, 2, 2), 5, "neumann", "p", "decrease by one"),
        ((8, 4, 4), 1, "dirichlet", "h", "bisect"),
        ((4, 4, 2), 3, ["neumann", "dirichlet", "dirichlet"], "h", "bisect"),
        ((4, 4, 4), 4, "dirichlet", "ph", "bisect"),
        # levels of >= 2^17 DoFs take the direction-by-direction form of the transfer
        ((16, 16, 12), 4, ["dirichlet", "neumann", "dirichlet"], "p", "bisect"),
        ((64, 32, 64), 1, ["neumann", "dirichlet", "dirichlet"], "h", "bisect"),
        ((12, 12, 12), 7, "dirichlet", "p", "decrease by one"),
    ],
)
def test_transfer_parity(pkg, nc, p, bc, mg_type, p_sequence):
    """prolongate_and_add / restrict_and_add of every level pair against the oracle's Kronecker-product transfer"""
    with pkg.Multigrid(fine_params(pkg, nc, p, bc), mg_type=mg_type, p_sequence=p_sequence, estimate_eigenvalues=False) as mg:
        seq = omg.level_sequence(nc, p, mg_type, p_sequence)
        assert mg.n_levels == len(seq)
        bcs = [BCMAP[b] for b in ([bc] * 3 if isinstance(bc, str) else bc)]
        meshes = [orc.Mesh(cells, deg, bcs) for cells, deg in seq]
        rng = np.random.default_rng(5)
        for l in range(1, mg.n_levels):
            T = omg.Transfer(meshes[l], meshes[l - 1])
            cf, cc = mg.levels[l], mg.levels[l - 1]
            assert cf.n_dofs == meshes[l].n_dofs and cc.n_dofs == meshes[l - 1].n_dofs
            xc, xf = rng.uniform(-1, 1, meshes[l - 1].n_dofs), rng.uniform(-1, 1, meshes[l].n_dofs)
            vc, vf = cc.vector(xc), cf.vector(xf)
            mg.prolongate_add(l, vf, vc)
            assert rel(vf.get(), xf + T.prolongate(xc)) < TOL
            rf = rng.uniform(-1, 1, meshes[l].n_dofs)
            wf, wc = cf.vector(rf), cc.vector(np.full(meshes[l - 1].n_dofs, 7.0))
            mg.restrict(l, wc, wf)
            assert rel(wc.get(), T.restrict(rf)) < TOL
            for v in (vc, vf, wf, wc):
                v.free()


@pytest.mark.parametrize(
    "nc,p,mg_type,bc",
    [
        ((20, 16, 12), 4, "ph", ["dirichlet", "neumann", "dirichlet"]),  # p = 4 -> 2 on the same cells
        ((40, 32, 24), 2, "h", ["dirichlet", "dirichlet", "neumann"]),   # h-coarsening at p = 2
        ((8, 8, 6), 7, "p", "dirichlet"),                                # p = 7 -> 3: widest table rows
        ((66, 34, 66), 1, "h", "dirichlet"),                             # ragged tiles (67 = 64 + 3 DoFs per row)
    ],
)
def test_transfer_forms_agree(pkg, monkeypatch, nc, p, mg_type, bc):
    """the direction-by-direction form (default on levels of >= 2^17 DoFs) and the single-kernel form
    (ASM_MG_DIRECT_TRANSFER) of the transfers give the same result (rounding apart)"""
    prm = fine_params(pkg, nc, p, bc)
    rng = np.random.default_rng(11)
    out = []
    for form in ("pass", "direct"):
        monkeypatch.delenv("ASM_MG_DIRECT_TRANSFER", raising=False)
        if form == "direct":
            monkeypatch.setenv("ASM_MG_DIRECT_TRANSFER", "1")
        with pkg.Multigrid(prm, mg_type=mg_type, estimate_eigenvalues=False) as mg:
            l = mg.n_levels - 1
            cf, cc = mg.levels[l], mg.levels[l - 1]
            assert cf.n_dofs >= 1 << 17
            if not out:
                xc, xf = rng.uniform(-1, 1, cc.n_dofs), rng.uniform(-1, 1, cf.n_dofs)
            vc, vf, wc = cc.vector(xc), cf.vector(xf), cc.vector(np.full(cc.n_dofs, 7.0))
            mg.restrict(l, wc, vf)
            mg.prolongate_add(l, vf, vc)
            out.append((vf.get(), wc.get()))
            for v in (vc, vf, wc):
                v.free()
    assert rel(out[0][0], out[1][0]) < TOL and rel(out[0][1], out[1][1]) < TOL


@pytest.mark.parametrize(
    "nc,p,bc,geometry,mg_type,weighting,degree,one_sided,poly",
    [
        ((4, 4, 4), 4, "dirichlet", "cartesian", "ph", "post", 2, False, "1st kind"),
        ((8, 4, 4), 4, "dirichlet", "cartesian", "ph", "symm", 1, True, "1st kind"),
        ((4, 4, 4), 3, "dirichlet", "cartesian", "h", "post", 3, False, "4th kind"),
        ((6, 2, 2), 4, "dirichlet", "kershaw", "p", "post", 2, False, "1st kind"),
    ],
)
def test_v_cycle_parity(pkg, nc, p, bc, geometry, mg_type, weighting, degree, one_sided, poly):
    """PreconditionMG::vmult: one V-cycle with Chebyshev + FDM smoothing on every level, coarse level = its smoother;
    the same eigenvalue bounds on both sides"""
    prm = fine_params(pkg, nc, p, bc, geometry, weighting=weighting, cheb_degree=degree, polynomial_type=poly)
    with pkg.Multigrid(prm, mg_type=mg_type, coarse="smoother", one_sided=one_sided, estimate_eigenvalues=False) as mg:
        lams = [1.3 + 0.1 * l for l in range(mg.n_levels)]
        for l, c in enumerate(mg.levels):
            c.set_eigenvalue(lams[l])
        M, meshes, ops = oracle_hierarchy(nc, p, bc, mg_type, weighting, 1, degree, lams, geometry, one_sided=one_sided,
                                          polynomial_type=poly)
        b = np.random.default_rng(9).uniform(-1, 1, meshes[-1].n_dofs)
        b[meshes[-1].constrained_mask()] = 0.0
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        mg.vmult(xv, bv)
        assert rel(xv.get(), M.vmult(b)) < 10 * TOL  # several sweeps deep: a few ulps more than one sweep
        bv.free()
        xv.free()


@pytest.mark.parametrize(
    "nc,p,bc,geometry,mg_type,solver,weighting,degree,rel_tol",
    [
        ((4, 4, 4), 4, "dirichlet", "cartesian", "ph", "GMRES", "post", 1, 1e-5),
        ((8, 4, 4), 4, "dirichlet", "cartesian", "ph", "CG", "symm", 2, 1e-6),
        ((4, 4, 4), 3, "dirichlet", "cartesian", "h", "GMRES", "post", 3, 1e-8),
        ((6, 4, 2), 4, "dirichlet", "kershaw", "p", "GMRES", "post", 2, 1e-5),
    ],
)
def test_multigrid_preconditioned_solve(pkg, nc, p, bc, geometry, mg_type, solver, weighting, degree, rel_tol):
    """solve() with the V-cycle as preconditioner and a CG coarse solve: iteration counts of the oracle twin, converged
    solution; both sides estimate their own eigenvalues (same algorithm, same start vector)"""
    prm = fine_params(pkg, nc, p, bc, geometry, weighting=weighting, cheb_degree=degree)
    with pkg.Multigrid(prm, mg_type=mg_type, coarse="cg", coarse_rel_tol=1e-4) as mg:
        M, meshes, ops = oracle_hierarchy(nc, p, bc, mg_type, weighting, 1, degree, None, geometry, coarse="cg", coarse_rel_tol=1e-4)
        for l in range(1, mg.n_levels):
            lam = mg.levels[l].cheb_parameters()["lambda_max"]
            assert abs(lam - M.smoothers[l].lambda_hat) < 1e-8 * lam
        mesh, op = meshes[-1], ops[-1]
        b = np.random.default_rng(2).uniform(0, 1, mesh.n_dofs)
        b[mesh.constrained_mask()] = 0.0
        fn = omg.solve_gmres if solver == "GMRES" else omg.solve_cg
        x_ref, it_ref = fn(op, b, M, rel_tol=rel_tol, abs_tol=0.0)
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        res = mg.fine.solve(xv, bv, solver=solver, preconditioner="Multigrid", mg=mg, rel_tol=rel_tol, abs_tol=0.0)
        assert res.converged == 1 and res.n_iterations == it_ref, (res.n_iterations, it_ref)
        x = xv.get()
        # The coarse CG makes the V-cycle a (slightly) non-linear preconditioner, so the residual estimate GMRES stops
        # on may differ from the true residual: the true residual is compared with the oracle twin's, which has the
        # same property, and agreement of the solutions is to solver accuracy, not to rounding.
        true_res = np.linalg.norm(b - op.vmult(x, unit_constrained=True))
        ref_res = np.linalg.norm(b - op.vmult(x_ref, unit_constrained=True))
        assert true_res <= 1.05 * ref_res + 1e-12 * np.linalg.norm(b)
        assert res.final_residual <= rel_tol * res.initial_residual
        assert rel(x, x_ref) < 1e-6
        bv.free()
        xv.free()


@pytest.mark.parametrize("solver", ["GMRES", "CG"])
@pytest.mark.parametrize("precond", ["Identity", "inner", "Chebyshev"])
def test_single_level_solvers(pkg, solver, precond):
    """the fixtures' other system preconditioners (Identity, FDM / Diagonal, Chebyshev) under both solvers, 3-D"""
    nc, p = (4, 3, 3), 3
    weighting = "symm"  # symmetric preconditioner: valid for CG too
    with pkg.Context(degree=p, n_cells=nc, boundary="dirichlet", weighting=weighting, cheb_degree=3, eig_n_iterations=20) as ctx:
        mesh = orc.Mesh(nc, p, [orc.DIRICHLET] * 3)
        op = orc.LaplaceOperator(mesh)
        fdm = orc.ASPoissonPreconditioner(mesh, 1, weighting)
        prec = {"Identity": omg.Identity(), "inner": fdm, "Chebyshev": orc.Chebyshev(op, fdm, degree=3, eig_n_iterations=20)}[precond]
        if precond == "Chebyshev":
            lam = ctx.estimate_eigenvalues()
            assert abs(lam - prec.lambda_hat) < 1e-9 * lam
        b = omg.right_hand_side_one(mesh)
        fn = omg.solve_gmres if solver == "GMRES" else omg.solve_cg
        x_ref, it_ref = fn(op, b, prec, rel_tol=1e-6, abs_tol=0.0)
        bv, xv = ctx.vector(b), ctx.vector(np.ones(mesh.n_dofs))  # x is reset to zero by the solver
        res = ctx.solve(xv, bv, solver=solver, preconditioner=precond, rel_tol=1e-6, abs_tol=0.0)
        assert res.converged == 1 and res.n_iterations == it_ref, (res.n_iterations, it_ref)
        assert rel(xv.get(), x_ref) < 1e-9
        assert abs(res.initial_residual - np.linalg.norm(b)) < 1e-12 * np.linalg.norm(b)


def test_gmres_restart_and_left_preconditioning(pkg):
    nc, p = (4, 4, 2), 2
    with pkg.Context(degree=p, n_cells=nc, boundary="dirichlet", precond="Diagonal", cheb_degree=1) as ctx:
        mesh = orc.Mesh(nc, p, [orc.DIRICHLET] * 3)
        op = orc.LaplaceOperator(mesh)
        jac = orc.JacobiPreconditioner(op)
        b = omg.right_hand_side_one(mesh)
        bv, xv = ctx.vector(b), ctx.vector()
        for right, tmp in [(True, 7), (False, 7), (False, 30)]:
            x_ref, it_ref = omg.solve_gmres(op, b, jac, rel_tol=1e-8, abs_tol=0.0, restart=tmp - 2, right=right)
            res = ctx.solve(xv, bv, solver="GMRES", preconditioner="inner", rel_tol=1e-8, abs_tol=0.0, right_preconditioning=right,
                            max_n_tmp_vectors=tmp)
            assert res.n_iterations == it_ref, (right, tmp, res.n_iterations, it_ref)
            assert rel(xv.get(), x_ref) < 1e-8


def test_multigrid_refuses_what_it_does_not_cover(pkg):
    with pytest.raises(RuntimeError, match="non-periodic"):
        pkg.Multigrid(fine_params(pkg, (4, 4, 4), 4, "periodic"), mg_type="p")
    with pytest.raises(RuntimeError, match="rank / world_size"):
        pkg.Multigrid(pkg.make_params(degree=2, n_cells=(4, 4, 4), world_size=2, rank=2))
    # (multi-rank hierarchies: tests/test_gpu_partitioned_mg.py)


@pytest.mark.parametrize(
    "nc,p,bc,geometry,mg_type,solver,weighting,degree",
    [
        ((8, 8, 4), 4, "dirichlet", "cartesian", "ph", "GMRES", "post", 1),
        ((4, 4, 4), 3, ["dirichlet", "neumann", "dirichlet"], "cartesian", "h", "CG", "symm", 2),
        ((6, 4, 2), 4, "dirichlet", "kershaw", "p", "GMRES", "post", 2),
    ],
)
def test_direct_coarse_solver(pkg, nc, p, bc, geometry, mg_type, solver, weighting, degree):
    """coarse-grid solver = dense inverse of the level operator: the V-cycle is linear and agrees with the oracle twin to
    rounding, and so do the iteration counts and the solution of a preconditioned solve"""
    prm = fine_params(pkg, nc, p, bc, geometry, weighting=weighting, cheb_degree=degree)
    with pkg.Multigrid(prm, mg_type=mg_type, coarse="direct") as mg:
        M, meshes, ops = oracle_hierarchy(nc, p, bc, mg_type, weighting, 1, degree, None, geometry, coarse="direct")
        mesh, op = meshes[-1], ops[-1]
        b = np.random.default_rng(3).uniform(0, 1, mesh.n_dofs)
        b[mesh.constrained_mask()] = 0.0
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        mg.vmult(xv, bv)
        assert rel(xv.get(), M.vmult(b)) < 1e-9  # eigenvalue estimates of both sides agree to ~1e-10 only
        fn = omg.solve_gmres if solver == "GMRES" else omg.solve_cg
        x_ref, it_ref = fn(op, b, M, rel_tol=1e-8, abs_tol=0.0)
        res = mg.fine.solve(xv, bv, solver=solver, preconditioner="Multigrid", mg=mg, rel_tol=1e-8, abs_tol=0.0)
        assert res.converged == 1 and res.n_iterations == it_ref, (res.n_iterations, it_ref)
        assert rel(xv.get(), x_ref) < 1e-7
        true_res = np.linalg.norm(b - op.vmult(xv.get(), unit_constrained=True))
        assert true_res <= 1.01e-8 * np.linalg.norm(b)  # linear preconditioner: the GMRES estimate is the true residual
        if solver == "GMRES":
            # keeping the preconditioned basis (one V-cycle less per cycle) gives the same iterates for a linear preconditioner
            x1 = xv.get()
            res2 = mg.fine.solve(xv, bv, solver=solver, preconditioner="Multigrid", mg=mg, rel_tol=1e-8, abs_tol=0.0,
                                 keep_preconditioned_basis=True)
            assert res2.n_iterations == res.n_iterations and rel(xv.get(), x1) < 1e-10
        bv.free()
        xv.free()


def test_direct_coarse_solver_refuses_large_levels(pkg):
    prm = fine_params(pkg, (13, 13, 13), 1, "dirichlet")
    with pytest.raises(Exception, match="direct coarse-grid solver"):
        pkg.Multigrid(prm, mg_type="h", coarse="direct")


@pytest.mark.parametrize("n_vec", [1, 3, 8, 11])
def test_gram_schmidt_sweeps(pkg, n_vec):
    """asm_vec_multi_dot / asm_vec_multi_axpy_norm (one Gram-Schmidt step of GMRES in two sweeps) against numpy"""
    with pkg.Context(fine_params(pkg, (6, 5, 4), 3, ["dirichlet", "neumann", "dirichlet"])) as c:
        rng = np.random.default_rng(17)
        wh = rng.standard_normal(c.n_dofs)
        vh = [rng.standard_normal(c.n_dofs) for _ in range(n_vec)]
        w, vs = c.vector(wh), [c.vector(v) for v in vh]
        dots = c.vec_multi_dot(w, vs)
        assert np.allclose(dots, [wh @ v for v in vh], rtol=1e-13, atol=1e-12)
        a = rng.standard_normal(n_vec)
        nn = c.vec_multi_axpy_norm(w, vs, a)
        ref = wh + sum(ai * v for ai, v in zip(a, vh))
        assert rel(w.get(), ref) < 1e-14
        assert abs(nn - ref @ ref) < 1e-12 * (ref @ ref)
        for v in [w] + vs:
            v.free()


# ---- the hierarchies of the BASELINE configurations 3, 4, 5 (experiments/large-scaling-opt_{fdm2, fdmv, ras}.json): their
#      smoothers inside the V-cycle and the preconditioned solve, 'ph' multigrid with the joint level kept twice and the
#      "mg intermediate smoother" (Chebyshev(4) over the diagonal) of those files on the degree-1 levels
CONFIG_HIERARCHIES = [
    # id, nc, p, geometry, weighting, n_overlap, element_centric, Chebyshev degree, solver, intermediate smoother
    ("config3_kershaw_p4_overlap2_post", (4, 4, 2), 4, "kershaw", "post", 2, True, 2, "GMRES", None),
    ("config4_vertex_star_symm_cg", (4, 4, 4), 4, "cartesian", "symm", 1, False, 2, "CG", ("Diagonal", 4)),
    ("config4_vertex_star_symm_kershaw", (4, 2, 2), 4, "kershaw", "symm", 1, False, 2, "CG", ("Diagonal", 4)),
    ("config5_kershaw_p7_ras_overlap2_k3", (4, 2, 2), 7, "kershaw", "ras", 2, True, 3, "GMRES", ("Diagonal", 4)),
    ("config5_cartesian_p7_ras_overlap2_k3", (4, 2, 2), 7, "cartesian", "ras", 2, True, 3, "GMRES", None),
]


@pytest.mark.parametrize("name,nc,p,geometry,weighting,n_overlap,element_centric,degree,solver,intermediate", CONFIG_HIERARCHIES,
                         ids=[c[0] for c in CONFIG_HIERARCHIES])
def test_v_cycle_and_solve_with_config_smoothers(pkg, name, nc, p, geometry, weighting, n_overlap, element_centric, degree, solver,
                                                 intermediate):
    """One V-cycle (fixed eigenvalue bounds on both sides) to 1e-11, and the preconditioned solve (direct coarse solver, own
    eigenvalue estimates) with the oracle twin's iteration count, for the smoothers of BASELINE configs 3-5."""
    prm = fine_params(pkg, nc, p, "dirichlet", geometry, weighting=weighting, n_overlap=n_overlap, element_centric=element_centric,
                      cheb_degree=degree)
    with pkg.Multigrid(prm, mg_type="ph", coarse="direct", estimate_eigenvalues=False, intermediate_smoother=intermediate) as mg:
        lams = [1.2 + 0.07 * l for l in range(mg.n_levels)]
        for l, c in enumerate(mg.levels):
            c.set_eigenvalue(lams[l])
        M, meshes, ops = oracle_hierarchy(nc, p, "dirichlet", "ph", weighting, n_overlap, degree, lams, geometry, coarse="direct",
                                          element_centric=element_centric, intermediate=intermediate)
        assert mg.n_levels == len(meshes)
        b = np.random.default_rng(21).uniform(-1, 1, meshes[-1].n_dofs)
        b[meshes[-1].constrained_mask()] = 0.0
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        mg.vmult(xv, bv)
        assert rel(xv.get(), M.vmult(b)) < 1e-11
        bv.free()
        xv.free()
    with pkg.Multigrid(prm, mg_type="ph", coarse="direct", intermediate_smoother=intermediate) as mg:
        M, meshes, ops = oracle_hierarchy(nc, p, "dirichlet", "ph", weighting, n_overlap, degree, None, geometry, coarse="direct",
                                          element_centric=element_centric, intermediate=intermediate)
        for l in range(1, mg.n_levels):
            lam = mg.levels[l].cheb_parameters()["lambda_max"]
            assert abs(lam - M.smoothers[l].lambda_hat) < 1e-7 * lam, (l, lam, M.smoothers[l].lambda_hat)
        mesh, op = meshes[-1], ops[-1]
        b = omg.right_hand_side_one(mesh)
        fn = omg.solve_gmres if solver == "GMRES" else omg.solve_cg
        x_ref, it_ref = fn(op, b, M, rel_tol=1e-6, abs_tol=0.0)
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        res = mg.fine.solve(xv, bv, solver=solver, preconditioner="Multigrid", mg=mg, rel_tol=1e-6, abs_tol=0.0)
        assert res.converged == 1 and res.n_iterations == it_ref, (res.n_iterations, it_ref)
        assert rel(xv.get(), x_ref) < 1e-6
        bv.free()
        xv.free()


@pytest.mark.parametrize("n_coarse_cycles,keep_duplicate", [(1, True), (2, True), (3, True), (1, False)])
def test_nested_hierarchies(pkg, n_coarse_cycles, keep_duplicate):
    """include/multigrid.h:260-375: the joint level of 'ph' twice (identity transfer), "mg intermediate smoother" on the
    lower multigrid, "n coarse cycles" Richardson steps of it as the coarse-grid solver of the upper one"""
    nc, p = (8, 4, 4), 4
    inter = ("Diagonal", 4)
    prm = fine_params(pkg, nc, p, "dirichlet", weighting="post", cheb_degree=2)
    with pkg.Multigrid(prm, mg_type="ph", coarse="direct", estimate_eigenvalues=False, intermediate_smoother=inter,
                       n_coarse_cycles=n_coarse_cycles, keep_duplicate_level=keep_duplicate) as mg:
        lams = [1.25 + 0.05 * l for l in range(mg.n_levels)]
        for l, c in enumerate(mg.levels):
            c.set_eigenvalue(lams[l])
        M, meshes, ops = oracle_hierarchy(nc, p, "dirichlet", "ph", "post", 1, 2, lams, coarse="direct", intermediate=inter,
                                          n_coarse_cycles=n_coarse_cycles, keep_duplicate=keep_duplicate)
        assert mg.n_levels == len(meshes) == (6 if keep_duplicate else 5)
        if keep_duplicate:
            # the two copies of the joint level: the transfer between them is the identity
            l = M.il
            assert meshes[l].n_dofs == meshes[l - 1].n_dofs
            v = np.random.default_rng(4).uniform(-1, 1, meshes[l].n_dofs)
            v[meshes[l].constrained_mask()] = 0.0
            vc, vf = mg.levels[l - 1].vector(v), mg.levels[l].vector()
            mg.prolongate_add(l, vf, vc)
            assert rel(vf.get(), v) < 1e-14
            mg.restrict(l, vc, vf)
            assert rel(vc.get(), v) < 1e-14
            vc.free()
            vf.free()
        b = np.random.default_rng(8).uniform(-1, 1, meshes[-1].n_dofs)
        b[meshes[-1].constrained_mask()] = 0.0
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        mg.vmult(xv, bv)
        assert rel(xv.get(), M.vmult(b)) < 1e-11
        bv.free()
        xv.free()
    # the solve with the levels' own eigenvalue estimates (the made-up bounds above need not give a contraction)
    with pkg.Multigrid(prm, mg_type="ph", coarse="direct", intermediate_smoother=inter, n_coarse_cycles=n_coarse_cycles,
                       keep_duplicate_level=keep_duplicate) as mg:
        M, meshes, ops = oracle_hierarchy(nc, p, "dirichlet", "ph", "post", 1, 2, None, coarse="direct", intermediate=inter,
                                          n_coarse_cycles=n_coarse_cycles, keep_duplicate=keep_duplicate)
        b = omg.right_hand_side_one(meshes[-1])
        x_ref, it_ref = omg.solve_gmres(ops[-1], b, M, rel_tol=1e-8, abs_tol=0.0)
        bv, xv = mg.fine.vector(b), mg.fine.vector()
        res = mg.fine.solve(xv, bv, solver="GMRES", preconditioner="Multigrid", mg=mg, rel_tol=1e-8, abs_tol=0.0)
        assert res.converged == 1 and res.n_iterations == it_ref and it_ref < 30, (res.n_iterations, it_ref)
        assert rel(xv.get(), x_ref) < 1e-7
        bv.free()
        xv.free()


@pytest.mark.parametrize("coarse,mg_type,n_coarse_cycles", [("smoother", "ph", 1), ("direct", "ph", 2), ("direct", "h", 1)])
def test_graph_of_small_levels(pkg, monkeypatch, coarse, mg_type, n_coarse_cycles):
    """hipGraph of the V-cycle's small levels (asm_mg_graph_replays): the first V-cycle runs eagerly, the second is
    captured, later ones replay - every one the V-cycle of a hierarchy without graphs to rounding (the default; ASM_MG_GRAPH
    switches them on), also
    after a level's Chebyshev parameters change (the graph is captured again)"""
    nc, p = (8, 8, 4), 4
    prm = fine_params(pkg, nc, p, "dirichlet", "cartesian", weighting="post", cheb_degree=2)
    kw = dict(mg_type=mg_type, coarse=coarse, estimate_eigenvalues=False, n_coarse_cycles=n_coarse_cycles)
    monkeypatch.delenv("ASM_MG_GRAPH", raising=False)
    plain = pkg.Multigrid(prm, **kw)
    monkeypatch.setenv("ASM_MG_GRAPH", "1")
    with plain, pkg.Multigrid(prm, **kw) as mg:
        assert plain.graph_replays() == (-1, 0)
        level, replays = mg.graph_replays()
        assert 1 <= level < mg.n_levels - 1 and replays == 0
        for h in (plain, mg):
            for l, c in enumerate(h.levels):
                c.set_eigenvalue(1.3 + 0.1 * l)
        rng = np.random.default_rng(4)
        vecs = [(h.fine.vector(), h.fine.vector()) for h in (plain, mg)]
        n = vecs[0][0].get().size
        for it in range(6):
            if it == 4:  # other parameters on a captured level: the next V-cycle captures again
                for h in (plain, mg):
                    h.levels[1].set_eigenvalue(2.0)
            b = rng.uniform(-1, 1, n)
            out = []
            for h, (xv, bv) in zip((plain, mg), vecs):
                bv.set(b)
                h.vmult(xv, bv)
                out.append(xv.get())
            assert rel(out[0], out[1]) < 1e-13, it  # (the low-order sweeps scatter with atomics: summation order, not bit for bit)
        assert mg.graph_replays() == (level, 5)  # every V-cycle but the first
        for xv, bv in vecs:
            xv.free()
            bv.free()
