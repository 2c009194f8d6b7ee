"""Partitioned (multi-rank) kernels, layouts and exchange plans on real device buffers.

RCCL cannot connect two ranks on one GPU, so all ranks of a partition live in this process as
separate contexts and the ghost exchange of MFRunner::loop (include/matrix_free_internal.h:
309-359) is carried out by the library's testing aid asm_debug_exchange_all (device-to-device
copies in place of ncclSend/ncclRecv, same pack/unpack kernels and exchange plan as the RCCL
path).  The partitioned result must equal the single-rank result, which test_gpu_parity.py
pins against the oracle."""
import numpy as np
import pytest

from conftest import load_package

pytestmark = pytest.mark.gpu
TOL = 1e-12


def owned_slices(layout, n_glob):
    b, s = layout.owned_begin, layout.owned_size
    return tuple(slice(b[d], b[d] + s[d]) for d in (2, 1, 0))


def run_partitioned(pkg, world, n_cells, boundary, kernel, what, weighting="post"):
    kw = dict(degree=4, n_cells=n_cells, boundary=boundary, precond="FDM", n_overlap=1, weighting=weighting, kernel=kernel)
    single = pkg.Context(**kw)
    npd = [4 * n_cells[d] + (0 if boundary == "periodic" else 1) for d in range(3)]
    rng = np.random.default_rng(7)
    xg = rng.standard_normal((npd[2], npd[1], npd[0]))
    assert xg.size == single.n_dofs
    apply_ = {"op": lambda c, d, s: c.op_vmult(d, s), "precond": lambda c, d, s: c.precond_vmult(d, s)}[what]
    xs, ys = single.vector(xg.reshape(-1)), single.vector()
    apply_(single, ys, xs)
    ref = ys.get().reshape(xg.shape)

    ranks = [pkg.Context(rank=r, world_size=world, **kw) for r in range(world)]
    assert all(c.kernel_variant() == single.kernel_variant() for c in ranks)
    src, dst = [], []
    for c in ranks:
        c.external_exchange(True)
        sl = owned_slices(c.layout, npd)
        src.append(c.vector(np.ascontiguousarray(xg[sl]).reshape(-1)))
        dst.append(c.vector())
    pkg.exchange_all(ranks, src, add=False)
    for c, d, s in zip(ranks, dst, src):
        apply_(c, d, s)
    if single.kernel_variant() == "generic":
        pkg.exchange_all(ranks, dst, add=True)
    out = np.full_like(ref, np.nan)
    for c, d in zip(ranks, dst):
        sl = owned_slices(c.layout, npd)
        out[sl] = d.get().reshape([c.layout.owned_size[k] for k in (2, 1, 0)])
    assert not np.isnan(out).any(), "owned ranges do not cover the global vector"
    err = np.abs(out - ref).max() / np.abs(ref).max()
    for c in ranks + [single]:
        c.close()
    return err


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("world,n_cells", [(2, (5, 4, 6)), (4, (4, 6, 6)), (8, (6, 6, 6))])
@pytest.mark.parametrize("boundary", ["dirichlet", "periodic"])
def test_tiled_partitioned(what, world, n_cells, boundary):
    pkg = load_package()
    assert run_partitioned(pkg, world, n_cells, boundary, "tiled", what) < TOL


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("world,n_cells", [(2, (4, 4, 6)), (8, (6, 6, 6))])
def test_generic_partitioned(what, world, n_cells):
    pkg = load_package()
    assert run_partitioned(pkg, world, n_cells, "dirichlet", "generic", what) < TOL


@pytest.mark.parametrize("weighting", ["symm", "none"])
def test_tiled_partitioned_weightings(weighting):
    pkg = load_package()
    assert run_partitioned(pkg, 4, (4, 6, 6), "dirichlet", "tiled", "precond", weighting) < TOL


def test_tiled_results_are_bitwise_reproducible():
    """No atomics and a fixed summation order in the tiled kernels."""
    pkg = load_package()
    ctx = pkg.Context(degree=4, n_cells=(13, 9, 7), boundary="dirichlet", precond="FDM", n_overlap=1, weighting="post", cheb_degree=3)
    assert ctx.kernel_variant() == "tiled"
    ctx.set_eigenvalue(1.7)
    rng = np.random.default_rng(3)
    b = ctx.vector(rng.standard_normal(ctx.n_owned))
    outs = []
    for _ in range(3):
        x = ctx.vector(np.zeros(ctx.n_owned))
        ctx.cheb_vmult(x, b)
        ctx.cheb_step(x, b)
        outs.append(x.get())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    ctx.close()
