"""The N > 1 path on CPU: two (and four) gloo ranks execute the library's ghost
exchange plans (asm_exchange_plan: exactly the plan asm_update_ghost_values /
asm_compress_add run over RCCL) on numpy boxes.  Every rank applies the oracle's
cell operation to ITS cells only, adds into its box (ghost planes included),
compresses, and the owned values must equal the serial oracle result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_plan(steps, box, layout, add):
    lo0 = [layout.store_begin[d] for d in range(3)]

    def view(lo, size):
        sl = tuple(slice(lo[d] - lo0[d], lo[d] - lo0[d] + size[d]) for d in (2, 1, 0))
        return box[sl]

    for st in steps:
        reqs, recv_bufs = [], {}
        for side in (0, 1):  # sends: low, then high
            if st.peer[side] >= 0 and np.prod(list(st.send_size[side])) > 0:
                buf = torch.from_numpy(np.ascontiguousarray(view(list(st.send_lo[side]), list(st.send_size[side]))))
                reqs.append(dist.isend(buf, st.peer[side]))
        for side in (1, 0):  # receives: high, then low
            if st.peer[side] >= 0 and np.prod(list(st.recv_size[side])) > 0:
                shape = tuple(st.recv_size[side][d] for d in (2, 1, 0))
                recv_bufs[side] = torch.empty(shape, dtype=torch.float64)
                reqs.append(dist.irecv(recv_bufs[side], st.peer[side]))
        for r in reqs:
            r.wait()
        for side, buf in recv_bufs.items():
            v = view(list(st.recv_lo[side]), list(st.recv_size[side]))
            if add:
                v += buf.numpy()
            else:
                v[...] = buf.numpy()
        if add:
            for side in (0, 1):
                if st.peer[side] >= 0 and np.prod(list(st.send_size[side])) > 0:
                    view(list(st.send_lo[side]), list(st.send_size[side]))[...] = 0.0


def _run_direct_plan(msgs, box, layout):
    """single-round update (asm_exchange_plan_direct): sends in array order, receives in reverse
    order, one tag: what RCCL's per-pair FIFO matching does"""
    lo0 = [layout.store_begin[d] for d in range(3)]

    def view(lo, size):
        sl = tuple(slice(lo[d] - lo0[d], lo[d] - lo0[d] + size[d]) for d in (2, 1, 0))
        return box[sl]

    reqs, recv_bufs = [], []
    for m in msgs:
        if m.send_count > 0:
            buf = torch.from_numpy(np.ascontiguousarray(view(list(m.send_lo), list(m.send_size))))
            reqs.append(dist.isend(buf, m.peer))
    for m in reversed(msgs):
        if m.recv_count > 0:
            buf = torch.empty(tuple(m.recv_size[d] for d in (2, 1, 0)), dtype=torch.float64)
            recv_bufs.append((m, buf))
            reqs.append(dist.irecv(buf, m.peer))
    for r in reqs:
        r.wait()
    for m, buf in recv_bufs:
        view(list(m.recv_lo), list(m.recv_size))[...] = buf.numpy()


def _run_direct_compress(msgs, box, layout):
    """single-round compress(add): the update's plan run backwards (every ghost box straight to its owner, added
    there, ghost boxes zeroed afterwards); sends in array order, receives in reverse order"""
    lo0 = [layout.store_begin[d] for d in range(3)]

    def view(lo, size):
        sl = tuple(slice(lo[d] - lo0[d], lo[d] - lo0[d] + size[d]) for d in (2, 1, 0))
        return box[sl]

    reqs, recv_bufs = [], []
    for m in msgs:
        if m.recv_count > 0:
            buf = torch.from_numpy(np.ascontiguousarray(view(list(m.recv_lo), list(m.recv_size))))
            reqs.append(dist.isend(buf, m.peer))
    for m in reversed(msgs):
        if m.send_count > 0:
            buf = torch.empty(tuple(m.send_size[d] for d in (2, 1, 0)), dtype=torch.float64)
            recv_bufs.append((m, buf))
            reqs.append(dist.irecv(buf, m.peer))
    for r in reqs:
        r.wait()
    for m, buf in recv_bufs:
        view(list(m.send_lo), list(m.send_size))[...] += buf.numpy()
    for m in msgs:
        if m.recv_count > 0:
            view(list(m.recv_lo), list(m.recv_size))[...] = 0.0


def _worker(rank, world, port, case, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_package

    pkg = load_package()
    import asm_oracle as orc

    nc, p, boundary, kw = case
    prm = pkg.make_params(degree=p, n_cells=nc, boundary=boundary, world_size=world, rank=rank, **kw)
    L = pkg.describe_layout(prm)
    bc = orc.PERIODIC if boundary == "periodic" else orc.DIRICHLET
    mesh = orc.Mesh(nc, p, [bc] * 3)
    n_overlap = kw.get("n_overlap", 1)
    elem = kw.get("element_centric", True)
    P = orc.ASPoissonPreconditioner(mesh, n_overlap, "post", elem)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, mesh.n_dofs)
    ref = P.vmult(x)

    nu = [mesh.Nu[d] for d in range(3)]
    glob = lambda a: a.reshape(nu[2], nu[1], nu[0])  # noqa: E731
    store = [L.store_size[d] for d in range(3)]
    lo0 = [L.store_begin[d] for d in range(3)]
    gb = [L.owned_begin[d] for d in range(3)]
    own = [L.owned_size[d] for d in range(3)]

    def unique_of(d, I_local):  # local logical coordinate -> global unique index
        I = gb[d] + I_local
        return I % nu[d] if boundary == "periodic" else I

    # ---- update_ghost_values -------------------------------------------------------
    box = np.full((store[2], store[1], store[0]), np.nan)
    osl = tuple(slice(-lo0[d], -lo0[d] + own[d]) for d in (2, 1, 0))
    gsl = tuple(slice(gb[d], gb[d] + own[d]) for d in (2, 1, 0))
    box[osl] = glob(x)[gsl]
    _run_plan(pkg.exchange_plan(prm, False), box, L, add=False)
    box2 = np.full((store[2], store[1], store[0]), np.nan)
    box2[osl] = glob(x)[gsl]
    _run_direct_plan(pkg.exchange_plan_direct(prm), box2, L)
    # both plans fill the same ghost entries with the same values
    ok_update = bool(np.array_equal(np.isnan(box), np.isnan(box2)) and np.array_equal(np.nan_to_num(box), np.nan_to_num(box2)))
    for k in range(store[2]):
        for j in range(store[1]):
            K, J = k + lo0[2], j + lo0[1]
            I = np.arange(store[0]) + lo0[0]
            gK, gJ, gI = unique_of(2, K), unique_of(1, J), np.array([unique_of(0, i) for i in I])
            inside = 0 <= gK < nu[2] and 0 <= gJ < nu[1]
            if not inside:
                continue
            m = (gI >= 0) & (gI < nu[0])
            want = glob(x)[gK, gJ, gI[m]]
            got = box[k, j, m]
            ok_update &= bool(np.array_equal(got, want))

    # ---- local cells scatter-add + compress(add) ------------------------------------
    T = np.where(P.valid, x[P.idx], 0.0)
    if P.w_pre is not None:
        T = T * P.w_pre
    T = P.apply_inverse(T)
    if P.w_post is not None:
        T = T * P.w_post
    acc = np.zeros((store[2], store[1], store[0]))
    cb = [L.cell_begin[d] for d in range(3)]
    cs = [L.cell_size[d] for d in range(3)]
    m1 = P.m
    o_add = (-(n_overlap - 1)) if elem else 1
    for c in range(mesh.ncells):
        cc = [mesh.cc[d][c] for d in range(3)]
        if not all(cb[d] <= cc[d] < cb[d] + cs[d] for d in range(3)):
            continue
        for k in range(m1):
            for j in range(m1):
                for i in range(m1):
                    if not P.valid[c, k, j, i]:
                        continue
                    loc = [p * (cc[d] - cb[d]) + o_add + (i, j, k)[d] for d in range(3)]
                    if boundary == "periodic":
                        for d in range(3):
                            if L.proc_grid[d] == 1:
                                loc[d] %= nc[d] * p
                    acc[loc[2] - lo0[2], loc[1] - lo0[1], loc[0] - lo0[0]] += T[c, k, j, i]
    acc2 = acc.copy()
    _run_plan(pkg.exchange_plan(prm, True), acc, L, add=True)
    _run_direct_compress(pkg.exchange_plan_direct(prm), acc2, L)
    got = acc[osl]
    want = glob(ref)[gsl]
    err = float(np.abs(got - want).max() / np.abs(ref).max())
    # the single-round form: same owned values (up to the order of the additions), ghost boxes zeroed
    err = max(err, float(np.abs(acc2[osl] - want).max() / np.abs(ref).max()))
    ghost = np.ones(acc2.shape, bool)
    ghost[osl] = False
    err = max(err, float(np.abs(acc2[ghost]).max()) if ghost.any() else 0.0)
    result_q.put((rank, ok_update, err))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ((4, 4, 4), 2, "dirichlet", dict(n_overlap=1, kernel="generic")),
    ((4, 4, 4), 2, "periodic", dict(n_overlap=1, kernel="generic")),
    ((4, 4, 4), 2, "dirichlet", dict(n_overlap=2, kernel="generic")),
    ((4, 4, 6), 2, "periodic", dict(element_centric=False, kernel="generic")),
    # the owner-computes kernels ghost p planes below every block instead of compressing
    ((3, 3, 4), 4, "periodic", dict(n_overlap=1, kernel="tiled")),
    ((3, 3, 4), 4, "dirichlet", dict(n_overlap=1, kernel="tiled")),
]


@pytest.mark.parametrize(
    "world,case",
    [(w, c) for w in (2, 4) for c in CASES] + [(8, CASES[1]), (8, CASES[4]), (8, CASES[5])],
    ids=lambda v: str(v) if isinstance(v, int) else f"{v[2]}-{v[3]}",
)
def test_exchange_plan_gloo(case, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + hash(str(case)) + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, ok_update, err in results:
        assert ok_update, f"rank {rank}: ghost values differ after update_ghost_values"
        assert err < 1e-13, f"rank {rank}: compress(add) result differs from the serial oracle by {err}"


# ---------------------------------------------------------------------------
# the exchange steps of the multi-rank multigrid transfers (csrc/multigrid.hip: asm_mg_restrict / asm_mg_prolongate_add):
# every rank sums over the fine DoFs it owns into its owned coarse planes and the coarse plane it shares with the next
# rank, compress(add) completes the shared planes at their owners; the prolongation reads the shared plane after
# update_ghost_values.  Executed here on numpy boxes over gloo with the library's own plans of the two levels.
# ---------------------------------------------------------------------------
def _mg_worker(rank, world, port, case, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_package

    pkg = load_package()
    import asm_oracle as orc
    import asm_oracle_mg as omg

    (ncf, pf), (ncc, pc) = case
    bcs = [orc.DIRICHLET] * 3
    mf, mc = orc.Mesh(ncf, pf, bcs), orc.Mesh(ncc, pc, bcs)
    T = omg.Transfer(mf, mc)
    P1 = [omg.interpolation_1d(mf, mc, d) for d in range(3)]  # [fine, coarse] per direction
    for P in P1:  # all faces Dirichlet: the first / last row and column belong to constrained DoFs (zero rows / columns)
        P[0, :] = P[-1, :] = 0.0
        P[:, 0] = P[:, -1] = 0.0
    rng = np.random.default_rng(3)
    rf, xc = rng.uniform(-1, 1, mf.n_dofs), rng.uniform(-1, 1, mc.n_dofs)
    ref_restrict, ref_prolong = T.restrict(rf), T.prolongate(xc)
    prm_f = pkg.make_params(degree=pf, n_cells=ncf, boundary="dirichlet", world_size=world, rank=rank, kernel="generic")
    Lf = pkg.describe_layout(prm_f)
    prm_c = pkg.make_params(degree=pc, n_cells=ncc, boundary="dirichlet", world_size=world, rank=rank, kernel="generic",
                            proc_grid=tuple(Lf.proc_grid))
    Lc = pkg.describe_layout(prm_c)
    nuf, nuc = [mf.Nu[d] for d in range(3)], [mc.Nu[d] for d in range(3)]
    gf = rf.reshape(nuf[2], nuf[1], nuf[0])
    gxc = xc.reshape(nuc[2], nuc[1], nuc[0])

    def ranges(L):
        own = [(L.owned_begin[d], L.owned_begin[d] + L.owned_size[d]) for d in range(3)]
        store = [(L.owned_begin[d] + L.store_begin[d], L.owned_begin[d] + L.store_begin[d] + L.store_size[d]) for d in range(3)]
        return own, store

    own_f, _ = ranges(Lf)
    own_c, store_c = ranges(Lc)
    # ---- restriction: partial sums over the owned fine DoFs, on the stored coarse box
    loc_f = gf[own_f[2][0]:own_f[2][1], own_f[1][0]:own_f[1][1], own_f[0][0]:own_f[0][1]]
    rows = [P1[d][own_f[d][0]:own_f[d][1], store_c[d][0]:store_c[d][1]] for d in range(3)]  # [fine owned, coarse stored]
    part = np.einsum("kji,kc,jb,ia->cba", loc_f, rows[2], rows[1], rows[0])
    # the library computes the owned planes and the shared top plane and zeroes every other ghost entry: the partial sums of
    # those are zero by themselves (interior coarse basis functions vanish on the faces of their cell)
    win = [(own_c[d][0], own_c[d][1] + (1 if Lc.proc_coord[d] + 1 < Lc.proc_grid[d] else 0)) for d in range(3)]
    outside = np.ones(part.shape, bool)
    outside[tuple(slice(win[d][0] - store_c[d][0], win[d][1] - store_c[d][0]) for d in (2, 1, 0))] = False
    zero_outside = float(np.abs(part[outside]).max()) if outside.any() else 0.0
    part[outside] = 0.0
    _run_direct_compress(pkg.exchange_plan_direct(prm_c), part, Lc)
    osl = tuple(slice(own_c[d][0] - store_c[d][0], own_c[d][1] - store_c[d][0]) for d in (2, 1, 0))
    gsl = tuple(slice(own_c[d][0], own_c[d][1]) for d in (2, 1, 0))
    err_r = float(np.abs(part[osl] - ref_restrict.reshape(nuc[2], nuc[1], nuc[0])[gsl]).max())
    # ---- prolongation: owned coarse values, update_ghost_values, then the owned fine rows
    box = np.full(part.shape, np.nan)
    box[osl] = gxc[gsl]
    _run_direct_plan(pkg.exchange_plan_direct(prm_c), box, Lc)
    box = np.nan_to_num(box, nan=1e30)  # an entry the rows of this rank read but nobody filled would show
    fine = np.einsum("cba,kc,jb,ia->kji", box, np.where(rows[2] != 0, rows[2], 0.0), np.where(rows[1] != 0, rows[1], 0.0),
                     np.where(rows[0] != 0, rows[0], 0.0))
    # (0 * 1e30 = 0: only entries with a non-zero weight count; a missing one gives ~1e30)
    fsl = tuple(slice(own_f[d][0], own_f[d][1]) for d in (2, 1, 0))
    err_p = float(np.abs(fine - ref_prolong.reshape(nuf[2], nuf[1], nuf[0])[fsl]).max())
    result_q.put((rank, err_r, err_p, zero_outside))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("case", [(((4, 4, 4), 4), ((4, 4, 4), 2)), (((8, 4, 4), 2), ((4, 2, 2), 2)), (((4, 4, 8), 3), ((4, 4, 8), 1))],
                         ids=["p4-p2", "h", "p3-p1"])
def test_multigrid_transfer_exchange_gloo(case, world):
    (ncf, _), (ncc, _) = case
    if ncf != ncc and world > 2:
        pytest.skip("the blocks of this level pair are nested for two ranks only")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + hash(str(case)) + world) % 2000
    procs = [ctx.Process(target=_mg_worker, args=(r, world, port, case, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, err_r, err_p, zero_outside in results:
        assert err_r < 1e-12, f"rank {rank}: restriction + compress(add) differs from the serial oracle by {err_r}"
        assert err_p < 1e-12, f"rank {rank}: update_ghost_values + prolongation differs from the serial oracle by {err_p}"
        assert zero_outside < 1e-14, f"rank {rank}: partial sums outside the restriction window are not zero ({zero_outside})"
