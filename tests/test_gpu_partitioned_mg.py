"""Multi-rank multigrid hierarchies (asm_mg_create with world_size > 1; include/multigrid.h:260-465 runs every level on the
same partition, element_centered_preconditioners_01.cc:540-740 builds the levels distributed).

RCCL cannot connect two ranks on one GPU: all ranks of a partition live in this process, every block-decomposed level gets a
loopback group (asm_debug_loopback_init: device-to-device copies in place of ncclSend / ncclRecv / ncclAllReduce, the same
plans, pack / unpack kernels and control flow as the RCCL path), and every rank is driven by its own host thread.  The
partitioned V-cycle and the preconditioned solve must reproduce the single-rank hierarchy, which tests/test_gpu_multigrid.py
pins against the oracle twin."""
import threading

import numpy as np
import pytest

from conftest import load_package
from test_gpu_partitioned import owned_slices

pytestmark = pytest.mark.gpu


def _run_ranks(fn, n):
    out = [None] * n

    def wrap(r):
        try:
            out[r] = fn(r)
        except Exception as e:  # noqa: BLE001 - reported by the caller
            out[r] = e

    threads = [threading.Thread(target=wrap, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "a rank thread hangs"
    for res in out:
        assert not isinstance(res, Exception), res
    return out


CASES = [
    # world, cells, degree, mg type, smoother weighting, n_overlap, intermediate smoother, geometry (None = Cartesian)
    (2, (8, 4, 4), 4, "ph", "post", 1, ("Diagonal", 4), None),
    (4, (8, 8, 4), 4, "ph", "post", 1, None, None),
    (8, (8, 8, 8), 4, "ph", "symm", 1, ("Diagonal", 3), None),
    (2, (12, 4, 4), 3, "h", "post", 1, None, None),
    (4, (6, 6, 4), 4, "p", "post", 2, None, None),
    (2, (8, 4, 4), 2, "hp", "post", 1, None, None),
    # BASELINE config 5's smoother on a partition: Kershaw mesh, p = 7, restricted additive Schwarz, n_overlap 2
    (2, (4, 4, 4), 7, "ph", "ras", 2, None, "kershaw"),
    (8, (4, 4, 4), 7, "p", "ras", 2, None, "kershaw"),
    # ... and config 3's (Kershaw, p = 4, n_overlap 2, post)
    (4, (8, 8, 4), 4, "ph", "post", 2, None, "kershaw"),
]


@pytest.mark.parametrize("world,nc,p,mg_type,weighting,n_overlap,inter,geometry", CASES)
def test_partitioned_hierarchy_reproduces_single_rank(world, nc, p, mg_type, weighting, n_overlap, inter, geometry):
    pkg = load_package()
    kw = dict(degree=p, n_cells=nc, boundary="dirichlet", weighting=weighting, n_overlap=n_overlap, cheb_degree=2)
    if geometry:
        kw.update(geometry=geometry, kershaw_eps=(0.3, 0.3))
    mkw = dict(mg_type=mg_type, coarse="direct", intermediate_smoother=inter)
    single = pkg.Multigrid(pkg.make_params(**kw), **mkw)
    npd = [p * nc[d] + 1 for d in range(3)]
    rng = np.random.default_rng(13)
    bg = rng.uniform(-1, 1, (npd[2], npd[1], npd[0]))
    bg[0, :, :] = bg[-1, :, :] = 0.0
    bg[:, 0, :] = bg[:, -1, :] = 0.0
    bg[:, :, 0] = bg[:, :, -1] = 0.0
    bs, xs = single.fine.vector(bg.reshape(-1)), single.fine.vector()
    single.vmult(xs, bs)
    x_ref = xs.get().reshape(bg.shape)
    res_ref = single.fine.solve(xs, bs, solver="GMRES", preconditioner="Multigrid", mg=single, rel_tol=1e-8, abs_tol=0.0)
    sol_ref = xs.get().reshape(bg.shape)
    lam_ref = [single.levels[l].cheb_parameters()["lambda_max"] for l in range(1, single.n_levels)]

    mgs = [pkg.Multigrid(pkg.make_params(rank=r, world_size=world, **kw), **mkw) for r in range(world)]
    assert all(m.n_levels == single.n_levels for m in mgs)
    dist = [mgs[0].level_distributed(l) for l in range(single.n_levels)]
    assert dist[-1] and not dist[0]  # the finest level is block-decomposed, the coarsest one replicated
    assert dist == sorted(dist)  # replicated levels below, block-decomposed levels above
    for l in range(single.n_levels):
        if dist[l]:
            pkg.loopback_init([m.levels[l] for m in mgs])

    def rank(r):
        m = mgs[r]
        m.after_loopback()
        c = m.fine
        sl = owned_slices(c.layout, npd)
        shape = [c.layout.owned_size[k] for k in (2, 1, 0)]
        bv, xv = c.vector(np.ascontiguousarray(bg[sl]).reshape(-1)), c.vector()
        m.vmult(xv, bv)
        x1 = xv.get().reshape(shape)
        res = c.solve(xv, bv, solver="GMRES", preconditioner="Multigrid", mg=m, rel_tol=1e-8, abs_tol=0.0)
        lam = [m.levels[l].cheb_parameters()["lambda_max"] for l in range(1, m.n_levels)]
        return sl, x1, xv.get().reshape(shape), res.n_iterations, res.converged, lam

    out = _run_ranks(rank, world)
    x_out, s_out = np.full_like(x_ref, np.nan), np.full_like(x_ref, np.nan)
    for sl, x1, x2, it, conv, lam in out:
        x_out[sl], s_out[sl] = x1, x2
        assert conv == 1 and it == res_ref.n_iterations, (it, res_ref.n_iterations)
        assert np.allclose(lam, lam_ref, rtol=1e-8), (lam, lam_ref)
    assert not np.isnan(x_out).any()
    assert np.abs(x_out - x_ref).max() < 1e-11 * np.abs(x_ref).max()
    assert np.abs(s_out - sol_ref).max() < 1e-7 * np.abs(sol_ref).max()
    for m in mgs + [single]:
        m.close()


def test_levels_below_the_nesting_limit_are_replicated():
    """16 x 4 x 4 cells on 2 ranks (split in z): 4 -> 2 cell layers keep the blocks nested (4 % (2 * 2) == 0), 2 -> 1 do not"""
    pkg = load_package()
    kw = dict(degree=1, n_cells=(16, 4, 4), boundary="dirichlet", weighting="post", cheb_degree=2)
    mgs = [pkg.Multigrid(pkg.make_params(rank=r, world_size=2, **kw), mg_type="h", coarse="direct", estimate_eigenvalues=False) for r in range(2)]
    m = mgs[0]
    cells = [tuple(m.levels[l].layout.cell_size) for l in range(m.n_levels)]
    dist = [m.level_distributed(l) for l in range(m.n_levels)]
    # levels (coarse -> fine): 4x1x1, 8x2x2, 16x4x4
    assert dist == [False, True, True], (dist, cells)
    for mm in mgs:
        mm.close()
