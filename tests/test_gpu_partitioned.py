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


def run_partitioned(pkg, world, n_cells, boundary, kernel, what, weighting="post", two_phase=False):
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
        c.two_phase(two_phase)
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


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("world,n_cells,boundary", [(2, (13, 9, 24), "dirichlet"), (8, (24, 16, 24), "dirichlet"), (4, (23, 30, 10), "periodic")])
def test_tiled_partitioned_two_phase(what, world, n_cells, boundary):
    """the launches of the overlapped exchange: workgroups next to interface faces first, the others
    second; together they must cover every workgroup exactly once"""
    pkg = load_package()
    assert run_partitioned(pkg, world, n_cells, boundary, "tiled", what, two_phase=True) < TOL


def _rank_thread(ctx, b_local, x0_local, keep, out, idx):
    try:
        ctx.set_keep_ghosts(keep)
        lam = ctx.estimate_eigenvalues()
        ctx.set_eigenvalue(1.8)
        bv, xv = ctx.vector(b_local), ctx.vector(x0_local)
        for _ in range(3):
            ctx.cheb_step(xv, bv)
        yv = ctx.vector()
        ctx.cheb_vmult(yv, bv)
        ctx.synchronize()
        out[idx] = (lam, xv.get(), yv.get())
    except Exception as e:  # noqa: BLE001
        out[idx] = e


@pytest.mark.parametrize("keep", [False, True])
@pytest.mark.parametrize("degree", [1, 3])
@pytest.mark.parametrize("world,n_cells,boundary", [(2, (13, 9, 24), "dirichlet"), (4, (12, 20, 16), "periodic"), (8, (24, 16, 24), "dirichlet")])
def test_multi_rank_control_flow_loopback(world, n_cells, boundary, degree, keep):
    """Eigenvalue estimate, repeated Chebyshev steps and a vmult with one host thread per rank and a
    loopback transport (asm_debug_loopback_init): comm stream, events, boundary / interior launches
    and the ghost bookkeeping of the overlapped exchange are the ones of the RCCL path."""
    import threading

    pkg = load_package()
    kw = dict(degree=4, n_cells=n_cells, boundary=boundary, precond="FDM", n_overlap=1, weighting="post", kernel="tiled", cheb_degree=degree)
    npd = [4 * n_cells[d] + (0 if boundary == "periodic" else 1) for d in range(3)]
    rng = np.random.default_rng(21)
    bg, xg = rng.standard_normal((npd[2], npd[1], npd[0])), rng.standard_normal((npd[2], npd[1], npd[0]))
    single = pkg.Context(**kw)
    lam_ref = single.estimate_eigenvalues()
    single.set_eigenvalue(1.8)
    bv, xv, yv = single.vector(bg.reshape(-1)), single.vector(xg.reshape(-1)), single.vector()
    for _ in range(3):
        single.cheb_step(xv, bv)
    single.cheb_vmult(yv, bv)
    x_ref, y_ref = xv.get().reshape(bg.shape), yv.get().reshape(bg.shape)

    ranks = [pkg.Context(rank=r, world_size=world, **kw) for r in range(world)]
    pkg.loopback_init(ranks)
    out = [None] * world
    threads = []
    for r, c in enumerate(ranks):
        sl = owned_slices(c.layout, npd)
        t = threading.Thread(target=_rank_thread, args=(c, np.ascontiguousarray(bg[sl]).reshape(-1), np.ascontiguousarray(xg[sl]).reshape(-1), keep, out, r))
        threads.append(t)
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a rank thread hangs"
    x_out, y_out = np.full_like(x_ref, np.nan), np.full_like(y_ref, np.nan)
    for c, res in zip(ranks, out):
        assert not isinstance(res, Exception), res
        lam, xl, yl = res
        assert abs(lam - lam_ref) < 1e-9 * lam_ref
        sl = owned_slices(c.layout, npd)
        shape = [c.layout.owned_size[k] for k in (2, 1, 0)]
        x_out[sl], y_out[sl] = xl.reshape(shape), yl.reshape(shape)
    assert np.abs(x_out - x_ref).max() < TOL * np.abs(x_ref).max()
    assert np.abs(y_out - y_ref).max() < TOL * np.abs(y_ref).max()
    for c in ranks + [single]:
        c.close()


@pytest.mark.parametrize("what", ["op", "precond"])
@pytest.mark.parametrize("n_overlap,element_centric", [(2, True), (1, False)])
def test_mixed_context_partitioned(what, n_overlap, element_centric):
    """degree 4 on Cartesian cells with 7^3 patches: tiled Laplace sweep + generic FDM sweep in one
    context, whose ghost layout (p planes below every block) serves both"""
    pkg = load_package()
    kw = dict(degree=4, n_cells=(6, 6, 8), boundary="dirichlet", precond="FDM", n_overlap=n_overlap, element_centric=element_centric,
              weighting="post")
    single = pkg.Context(**kw)
    assert single.kernel_variant() == "tiled"
    npd = [4 * 6 + 1, 4 * 6 + 1, 4 * 8 + 1]
    rng = np.random.default_rng(17)
    xg = rng.standard_normal((npd[2], npd[1], npd[0]))
    apply_ = {"op": lambda c, d, s: c.op_vmult(d, s), "precond": lambda c, d, s: c.precond_vmult(d, s)}[what]
    xs, ys = single.vector(xg.reshape(-1)), single.vector()
    apply_(single, ys, xs)
    ref = ys.get().reshape(xg.shape)
    world = 4
    ranks = [pkg.Context(rank=r, world_size=world, **kw) for r in range(world)]
    src, dst = [], []
    for c in ranks:
        c.external_exchange(True)
        sl = owned_slices(c.layout, npd)
        src.append(c.vector(np.ascontiguousarray(xg[sl]).reshape(-1)))
        dst.append(c.vector())
    pkg.exchange_all(ranks, src, add=False)
    for c, d, s in zip(ranks, dst, src):
        apply_(c, d, s)
    if what == "precond":  # the generic FDM sweep leaves contributions in the ghost planes
        pkg.exchange_all(ranks, dst, add=True)
    out = np.full_like(ref, np.nan)
    for c, d in zip(ranks, dst):
        sl = owned_slices(c.layout, npd)
        out[sl] = d.get().reshape([c.layout.owned_size[k] for k in (2, 1, 0)])
    assert np.abs(out - ref).max() < TOL * np.abs(ref).max()
    for c in ranks + [single]:
        c.close()
