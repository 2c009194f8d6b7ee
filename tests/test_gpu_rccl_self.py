"""The multi-rank path on real hardware through RCCL, on the one GPU a test box has.

RCCL refuses two ranks on one device, so a genuine 2-rank run cannot happen here.  With ASM_SELF_INTERFACE the single rank
closes its periodic directions through ghost planes and messages to ITSELF instead of index wrap-around: the context is
then a "multi-rank" one in every respect but the peer (26-neighbour plan, pack/unpack kernels, ncclGroupStart / ncclSend /
ncclRecv / ncclGroupEnd on the communication stream, the event choreography of the overlapped exchange, single-round
compress(add), ncclAllReduce in the dot products).  Results must equal those of the ordinary wrap-around context, which
test_gpu_parity.py pins against the oracle (MFRunner::loop, include/matrix_free_internal.h:309-359)."""
import numpy as np
import pytest

from conftest import load_package

pytestmark = pytest.mark.gpu
TOL = 1e-12


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


CASES = [
    # degree, cells, boundary, precond kw, kernel
    (4, (6, 4, 6), "periodic", dict(precond="FDM", n_overlap=1, weighting="post"), "tiled"),
    (4, (4, 6, 4), ["periodic", "dirichlet", "periodic"], dict(precond="FDM", n_overlap=1, weighting="symm"), "tiled"),
    (3, (4, 4, 6), "periodic", dict(precond="FDM", element_centric=False, weighting="post"), "generic"),
    (4, (4, 4, 4), ["dirichlet", "periodic", "neumann"], dict(precond="FDM", n_overlap=2, weighting="post"), "generic"),
    (2, (6, 6, 4), "periodic", dict(precond="Diagonal"), "generic"),
]


@pytest.mark.parametrize("degree,n_cells,boundary,pkw,kernel", CASES)
def test_self_interface_through_rccl(monkeypatch, degree, n_cells, boundary, pkw, kernel):
    pkg = load_package()
    kw = dict(degree=degree, n_cells=n_cells, boundary=boundary, kernel=kernel, cheb_degree=3, **pkw)
    wrap = pkg.Context(**kw)
    monkeypatch.setenv("ASM_SELF_INTERFACE", "1")
    own = pkg.Context(**kw)
    monkeypatch.delenv("ASM_SELF_INTERFACE")
    assert own.n_dofs == wrap.n_dofs and own.layout.n_ghost > 0 and wrap.layout.n_ghost == 0
    own.comm_init(pkg.comm_unique_id())
    rng = np.random.default_rng(21)
    x, b = rng.standard_normal(wrap.n_dofs), rng.standard_normal(wrap.n_dofs)
    out = []
    for c in (wrap, own):
        xs, bs, y = c.vector(x), c.vector(b), c.vector()
        c.constraints_set_zero(xs)
        c.constraints_set_zero(bs)
        res = {}
        c.op_vmult(y, xs)
        res["op"] = y.get()
        c.precond_vmult(y, bs)
        res["precond"] = y.get()
        res["dot"] = c.vec_dot(xs, bs)
        c.set_eigenvalue(1.7)
        c.cheb_vmult(y, bs)
        res["cheb_vmult"] = y.get()
        c.cheb_step(xs, bs)
        res["cheb_step"] = xs.get()
        res["lambda"] = c.estimate_eigenvalues()
        out.append(res)
        for v in (xs, bs, y):
            v.free()
    for key in ("op", "precond", "cheb_vmult", "cheb_step"):
        assert rel(out[1][key], out[0][key]) < TOL, key
    assert abs(out[1]["dot"] - out[0]["dot"]) < 1e-12 * abs(out[0]["dot"]) + 1e-12
    assert abs(out[1]["lambda"] - out[0]["lambda"]) < 1e-8 * out[0]["lambda"]
    own.close()
    wrap.close()
