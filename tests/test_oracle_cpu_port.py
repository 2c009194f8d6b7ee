"""The threaded C++ CPU restatement (oracle/asm_oracle_cpu.cc, used as cpu_baseline) agrees
with the numpy oracle."""
import os
import subprocess

import numpy as np
import pytest

import asm_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cpu_mod():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    import asm_oracle_cpu

    return asm_oracle_cpu


@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("degree", [1, 3])
@pytest.mark.parametrize("threads", [1, 3])
def test_cpu_port_matches_numpy_oracle(cpu_mod, periodic, degree, threads):
    nc = (4, 6, 4)
    extent = (1.0, 1.5, 0.5)
    bc = orc.PERIODIC if periodic else orc.DIRICHLET
    mesh = orc.Mesh(nc, 4, [bc] * 3, h=[extent[d] / nc[d] for d in range(3)])
    op = orc.LaplaceOperator(mesh)
    prec = orc.ASPoissonPreconditioner(mesh, 1, "post")
    cheb = orc.Chebyshev(op, prec, degree=degree, lambda_max=1.8)
    rng = np.random.default_rng(4)
    x0, b = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
    sm = cpu_mod.CpuSmoother(nc, periodic, extent, degree=degree, lambda_max=1.8, n_threads=threads)
    assert sm.n_dofs == mesh.n_dofs
    got = sm.step(x0, b)
    ref = cheb.step(x0, b)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-12


@pytest.mark.parametrize("degree", [1, 2])
@pytest.mark.parametrize("threads", [1, 3])
def test_cpu_port_deformed_matches_numpy_oracle(cpu_mod, degree, threads):
    """the threaded deformed-mesh leg (cpu_baseline of the Kershaw workload) against the numpy oracle"""
    nc = (6, 4, 2)
    mesh = orc.Mesh(nc, 4, [orc.DIRICHLET] * 3, mapfn=lambda pts: orc.kershaw(0.3, 0.3, pts))
    op = orc.LaplaceOperator(mesh)
    prec = orc.ASPoissonPreconditioner(mesh, 1, "post")
    cheb = orc.Chebyshev(op, prec, degree=degree, lambda_max=1.8)
    rng = np.random.default_rng(4)
    x0, b = rng.uniform(-1, 1, mesh.n_dofs), rng.uniform(-1, 1, mesh.n_dofs)
    sm = cpu_mod.CpuSmootherDeformed(mesh, degree=degree, lambda_max=1.8, n_threads=threads)
    got = sm.step(x0, b)
    ref = cheb.step(x0, b)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-12


@pytest.mark.parametrize("n_overlap", [1, 2])
@pytest.mark.parametrize("bc", ["dirichlet", "mixed"])
def test_batched_eigen_decompositions(n_overlap, bc):
    """the batched eigen-decompositions (large deformed meshes) give the same FDM inverse as the loop over cells"""
    nc = (6, 3, 2)
    bcs = [orc.DIRICHLET] * 3 if bc == "dirichlet" else [orc.DIRICHLET, orc.NEUMANN, orc.DIRICHLET]
    mesh = orc.Mesh(nc, 4, bcs, mapfn=lambda pts: orc.kershaw(0.3, 0.3, pts))
    a = orc.ASPoissonPreconditioner(mesh, n_overlap, "post", batched_eigen=False)
    b = orc.ASPoissonPreconditioner(mesh, n_overlap, "post", batched_eigen=True)
    v = np.random.default_rng(6).uniform(-1, 1, mesh.n_dofs)
    ra, rb = a.vmult(v), b.vmult(v)
    assert np.linalg.norm(ra - rb) < 1e-12 * np.linalg.norm(ra)
    assert np.allclose(np.sort(a.lam, axis=-1), np.sort(b.lam, axis=-1), rtol=1e-10, atol=1e-9)
