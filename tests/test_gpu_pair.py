"""GPU parity of the wave-pair sweeps (dealii-asm_amd/csrc/kernels_pair.hip): the degree-4 Laplace operator on box cells
and the FDM preconditioner with cell patches on Cartesian meshes, forced by ASM_PAIR=1 (a context otherwise measures them
against the tiled kernels and takes the faster family, so the other test modules may never run them).  Same oracle and
tolerance as test_gpu_parity.py: relative l2 difference < 1e-12 (BASELINE.json north_star)."""
import numpy as np
import pytest

import asm_oracle as orc
from test_gpu_parity import make_pair, rel
from test_gpu_partitioned import run_partitioned

pytestmark = pytest.mark.gpu
TOL = 1e-12

# tiles of 8 x 8 cells own 7 x 7 (tile 0 at a physical boundary: all 8); the cell behind the last one owns the last DoF
# plane: one tile, two tiles, a last tile that holds only that cell (7 + 1, 14 + 1 cells), periodic blocks (halo tile
# column), mixed boundaries, several z chunks, z classes changing inside a chunk
PAIR_CASES = [
    ((3, 2, 2), "periodic"),
    ((3, 3, 2), "dirichlet"),
    ((7, 8, 3), "dirichlet"),
    ((8, 7, 2), ["neumann", "dirichlet", "periodic"]),
    ((9, 15, 4), "dirichlet"),
    ((14, 9, 2), "neumann"),
    ((17, 4, 3), ["dirichlet", "periodic", "neumann"]),
    ((4, 16, 5), ["periodic", "dirichlet", "dirichlet"]),
    ((16, 16, 9), "periodic"),
    ((23, 15, 2), "dirichlet"),
    ((1, 1, 1), "dirichlet"),
    ((2, 1, 3), "neumann"),
]


@pytest.fixture()
def force_pair(monkeypatch):
    monkeypatch.setenv("ASM_PAIR", "1")


@pytest.mark.parametrize("weighting", ["none", "pre", "post", "symm", "ras"])
@pytest.mark.parametrize("nc,bc", PAIR_CASES)
def test_precond_vmult_pair(pkg, force_pair, nc, bc, weighting):
    mesh, ctx = make_pair(pkg, nc, 4, bc, kernel="tiled", weighting=weighting)
    with ctx:
        assert ctx.pair_sweeps() == (True, True)
        rng = np.random.default_rng(7)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        ref = orc.ASPoissonPreconditioner(mesh, 1, weighting).vmult(src)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 123.0))  # dst needs no zeroing
        ctx.precond_vmult(y, x)
        assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("nc,bc", PAIR_CASES)
def test_op_vmult_pair(pkg, force_pair, nc, bc):
    mesh, ctx = make_pair(pkg, nc, 4, bc, kernel="tiled", extent=(1.0, 0.7, 1.3))
    with ctx:
        assert ctx.pair_sweeps()[0]
        rng = np.random.default_rng(3)
        src = rng.uniform(-1, 1, mesh.n_dofs)
        x, y = ctx.vector(src), ctx.vector(np.full(mesh.n_dofs, 5.0))
        ctx.op_vmult(y, x)
        assert rel(y.get(), orc.LaplaceOperator(mesh).vmult(src)) < TOL


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
@pytest.mark.parametrize("nc,bc", PAIR_CASES[:10])
def test_chebyshev_pair(pkg, force_pair, nc, bc, degree):
    """every fused update of both sweeps: residual, first sweep of a zero start, two- and three-term updates"""
    mesh, ctx = make_pair(pkg, nc, 4, bc, kernel="tiled", weighting="post", cheb_degree=degree)
    with ctx:
        assert ctx.pair_sweeps() == (True, True)
        op = orc.LaplaceOperator(mesh)
        cheb = orc.Chebyshev(op, orc.ASPoissonPreconditioner(mesh, 1, "post"), degree=degree, lambda_max=1.9)
        ctx.set_eigenvalue(1.9)
        rng = np.random.default_rng(11)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector()
        ctx.cheb_vmult(xv, bv)
        assert rel(xv.get(), cheb.vmult(b)) < TOL
        xv.set(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


def test_pair_not_used_where_it_does_not_apply(pkg, force_pair):
    """deformed meshes keep the tiled FDM sweep and the generic Laplace sweep; other degrees the generic kernels"""
    with pkg.Context(degree=4, n_cells=(6, 2, 2), boundary="dirichlet", geometry="kershaw", kershaw_eps=(0.3, 0.3)) as ctx:
        assert ctx.pair_sweeps() == (False, False)
    with pkg.Context(degree=3, n_cells=(4, 4, 4), boundary="dirichlet") as ctx:
        assert ctx.pair_sweeps() == (False, False)


@pytest.mark.parametrize("nc,periodic,degree", [((80, 80, 80), False, 1), ((80, 80, 80), False, 3), ((64, 32, 32), True, 2), ((128, 128, 96), False, 1)])
def test_full_size_pair_equals_tiled(pkg, monkeypatch, nc, periodic, degree):
    """the bench workload at full size: the wave-pair kernels against the tiled kernels (which
    test_gpu_parity_configs.py::test_full_size_bench_workload pins on the threaded C++ restatement)"""
    rng = np.random.default_rng(5)
    out = []
    for pair in ("0", "1"):
        monkeypatch.setenv("ASM_PAIR", pair)
        with pkg.Context(degree=4, n_cells=nc, boundary="periodic" if periodic else "dirichlet", cheb_degree=degree, kernel="tiled") as ctx:
            assert ctx.pair_sweeps() == ((pair == "1"),) * 2
            if not out:
                b, x0 = rng.uniform(-1, 1, ctx.n_dofs), rng.uniform(-1, 1, ctx.n_dofs)
            ctx.set_eigenvalue(1.7)
            bv, xv = ctx.vector(b), ctx.vector(x0)
            ctx.cheb_step(xv, bv)
            out.append(xv.get())
    assert rel(out[1], out[0]) < TOL


@pytest.mark.parametrize("waves", [2, 4])
@pytest.mark.parametrize("zchunks", [1, 2, 5, 64])
@pytest.mark.parametrize("nc,bc", [((9, 15, 5), "dirichlet"), ((16, 9, 9), "periodic"), ((8, 7, 6), ["neumann", "dirichlet", "periodic"])])
def test_pair_forms_and_chunkings(pkg, force_pair, monkeypatch, nc, bc, waves, zchunks):
    """both forms of the kernel (two / four waves per tile; a context otherwise measures and picks one) with pinned z
    chunkings from one chunk to one layer per chunk: every fused update through a Chebyshev step of degree 3"""
    monkeypatch.setenv("ASM_PAIR_WAVES", str(waves))
    monkeypatch.setenv("ASM_PAIR_ZCHUNKS", str(zchunks))
    mesh, ctx = make_pair(pkg, nc, 4, bc, kernel="tiled", weighting="symm", cheb_degree=3)
    with ctx:
        assert ctx.pair_sweeps() == (True, True) and ctx.pair_waves() == waves and ctx.pair_zchunks() == zchunks
        plan, _ = pkg.pair_plan(ctx.params, 256)  # (ASM_PAIR_ZCHUNKS applies to the plan as well)
        assert plan["nzc"] <= min(zchunks, nc[2]) and plan["nzc"] * plan["zlen"] >= nc[2] > (plan["nzc"] - 1) * plan["zlen"]
        cheb = orc.Chebyshev(orc.LaplaceOperator(mesh), orc.ASPoissonPreconditioner(mesh, 1, "symm"), degree=3, lambda_max=1.8)
        ctx.set_eigenvalue(1.8)
        rng = np.random.default_rng(13)
        b, x0 = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
        bv, xv = ctx.vector(b), ctx.vector(x0)
        ctx.cheb_step(xv, bv)
        assert rel(xv.get(), cheb.step(x0, b)) < TOL


def test_measured_pick_is_consistent(pkg, monkeypatch):
    """without ASM_PAIR a context measures chunk counts, forms and families when it is created: whatever it picks, the step
    equals the tiled kernels' (and a second context of the same shape takes the same pick from the cache)"""
    for v in ("ASM_PAIR", "ASM_PAIR_WAVES", "ASM_PAIR_ZCHUNKS"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(17)
    picks, out = [], []
    for rep in range(3):
        if rep == 2:
            monkeypatch.setenv("ASM_PAIR", "0")
        with pkg.Context(degree=4, n_cells=(24, 16, 12), boundary="dirichlet", cheb_degree=2, kernel="tiled") as ctx:
            picks.append((ctx.pair_sweeps(), ctx.pair_zchunks(), ctx.pair_waves()))
            if not out:
                b, x0 = rng.uniform(-1, 1, ctx.n_dofs), rng.uniform(-1, 1, ctx.n_dofs)
            ctx.set_eigenvalue(1.7)
            bv, xv = ctx.vector(b), ctx.vector(x0)
            ctx.cheb_step(xv, bv)
            out.append(xv.get())
    assert picks[0] == picks[1] and picks[2][0] == (False, False)
    assert rel(out[0], out[2]) < TOL and rel(out[1], out[2]) < TOL


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("world,n_cells", [(2, (5, 4, 6)), (4, (4, 6, 6)), (8, (6, 6, 6))])
@pytest.mark.parametrize("boundary", ["dirichlet", "periodic"])
def test_pair_partitioned(pkg, force_pair, what, world, n_cells, boundary):
    """blocks with ghost planes and halo cells of lower ranks: the partitioned result equals the single-rank one"""
    assert run_partitioned(pkg, world, n_cells, boundary, "tiled", what) < TOL


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("world,n_cells", [(2, (9, 4, 6)), (4, (4, 9, 6)), (8, (6, 6, 6))])
@pytest.mark.parametrize("boundary", ["dirichlet", "periodic"])
def test_pair_partitioned_two_phase(pkg, force_pair, what, world, n_cells, boundary):
    """two-phase launches of the pair kernels (the workgroups at rank interfaces first, as with an overlapped exchange)"""
    assert run_partitioned(pkg, world, n_cells, boundary, "tiled", what, two_phase=True) < TOL


def test_pair_equals_tiled_on_random_blocks(pkg, monkeypatch):
    """twenty blocks of random extents (all leftovers of the 7-cell tiling: plain last tiles and strip tiles) and boundary
    kinds: a Chebyshev step of degree 2 with the pair kernels (both forms) against the tiled kernels"""
    rng = np.random.default_rng(2024)
    kinds = ["dirichlet", "neumann", "periodic"]
    for case in range(20):
        nc = tuple(int(v) for v in rng.integers(1, 40, 2)) + (int(rng.integers(1, 9)),)
        bc = [kinds[int(k)] for k in rng.integers(0, 3, 3)]
        b = x0 = None
        out = {}
        for pair, waves in (("0", "4"), ("1", "4"), ("1", "2")):
            monkeypatch.setenv("ASM_PAIR", pair)
            monkeypatch.setenv("ASM_PAIR_WAVES", waves)
            with pkg.Context(degree=4, n_cells=nc, boundary=bc, cheb_degree=2, kernel="tiled", weighting="symm") as ctx:
                if b is None:
                    b, x0 = rng.uniform(-1, 1, ctx.n_dofs), rng.uniform(-1, 1, ctx.n_dofs)
                ctx.set_eigenvalue(1.6)
                bv, xv = ctx.vector(b), ctx.vector(x0)
                ctx.cheb_step(xv, bv)
                out[(pair, waves)] = xv.get()
        for key in (("1", "4"), ("1", "2")):
            assert rel(out[key], out[("0", "4")]) < TOL, (case, nc, bc, key)
