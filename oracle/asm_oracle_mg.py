"""Oracle twin (numpy, TEST INFRASTRUCTURE ONLY) of the callers of the smoother: transfer operators,
the multigrid V-cycle, the coarse solver and the outer Krylov solvers.

Restated from
  include/multigrid.h:260-465 (MyMultigrid::do_update / vmult: deal.II Multigrid + PreconditionMG),
  element_centered_preconditioners_01.cc:108-263 (solve: SolverCG / SolverGMRES with ReductionControl),
  element_centered_preconditioners_01.cc:540-740 (level construction: "mg type" h | p | hp | ph,
  "mg p sequence" bisect | decrease by one | go to one),
  transfer_01.likwid.cc:21-137 (MGTwoLevelTransfer between two operators);
the deal.II pieces they call (Multigrid::level_v_step, MGTransferGlobalCoarsening, SolverGMRES with right
preconditioning, SolverCG, MGSmootherRelaxation) are restated from their mathematics.

Pinned by tests/test_oracle_mg_golden.py on the iteration counts printed in the reference's own
tests/element_centered_preconitioners/small/*.output (2-D, p=3, 8x8 cells, GMRES to 1e-2)."""
from __future__ import annotations

import numpy as np

import asm_oracle as orc


# --------------------------------------------------------------------------
# transfer: both spaces are tensor-product spaces on nested structured meshes, so the global
# prolongation matrix is the Kronecker product of 1-D interpolation matrices
# (MGTransferGlobalCoarsening: cell-wise interpolation, weighted by 1/valence on shared DoFs, which for
# continuous functions is the nodal interpolation restated here; restriction is its transpose)
# --------------------------------------------------------------------------
def interpolation_1d(mesh_f: orc.Mesh, mesh_c: orc.Mesh, d: int) -> np.ndarray:
    """P[I_f, J_c] = coarse global 1-D basis function J_c at fine node I_f (direction d)."""
    pf, pc = mesh_f.p, mesh_c.p
    ncf, ncc = mesh_f.nc[d], mesh_c.nc[d]
    assert not mesh_f.periodic[d], "multigrid twin: non-periodic directions only"
    if ncf == ncc:
        ratio = 1
    else:
        assert ncf == 2 * ncc and pf == pc, "h-coarsening halves the cells at equal degree"
        ratio = 2
    nodes_f, nodes_c = mesh_f.basis.nodes, mesh_c.basis.nodes
    P = np.zeros((mesh_f.Nu[d], mesh_c.Nu[d]))
    for cf in range(ncf):
        cc, s = cf // ratio, cf % ratio
        xi = (s + nodes_f) / ratio  # fine nodes of this cell in the coarse cell's reference coordinate
        V, _ = orc.lagrange_eval(nodes_c, xi)
        P[pf * cf : pf * cf + pf + 1, pc * cc : pc * cc + pc + 1] = V
    return P


class Transfer:
    """MGTwoLevelTransfer between two levels (prolongate / restrict_and_add of include/multigrid.h:289-299,
    336-346); constrained DoFs of either level carry zeros."""

    def __init__(self, mesh_f: orc.Mesh, mesh_c: orc.Mesh):
        self.mf, self.mc = mesh_f, mesh_c
        self.P = [interpolation_1d(mesh_f, mesh_c, d) for d in range(mesh_f.dim)]
        self.con_f, self.con_c = mesh_f.constrained_mask(), mesh_c.constrained_mask()

    def _apply(self, mats, x, shape_in):
        dim = len(mats)
        T = x.reshape(shape_in[::-1])
        for d in range(dim):
            ax = dim - 1 - d
            T = np.moveaxis(np.tensordot(mats[d], T, axes=([1], [ax])), 0, ax)
        return T.reshape(-1)

    def prolongate(self, xc):
        xc = np.where(self.con_c, 0.0, xc)
        xf = self._apply(self.P, xc, self.mc.Nu)
        xf[self.con_f] = 0.0
        return xf

    def restrict(self, rf):
        rf = np.where(self.con_f, 0.0, rf)
        rc = self._apply([P.T for P in self.P], rf, self.mf.Nu)
        rc[self.con_c] = 0.0
        return rc


# --------------------------------------------------------------------------
# level hierarchy
# --------------------------------------------------------------------------
def polynomial_coarsening_sequence(p, kind="bisect"):
    """MGTransferGlobalCoarseningTools::create_polynomial_coarsening_sequence [deal.II]: ascending degrees"""
    seq = [p]
    while seq[-1] > 1:
        q = seq[-1]
        seq.append({"bisect": max(q // 2, 1), "decrease by one": q - 1, "go to one": 1}[kind])
    return seq[::-1]


def geometric_coarsening_sequence(nc):
    """cell counts of create_geometric_coarsening_sequence for a structured box: halve while all are even"""
    seq = [tuple(nc)]
    while all(n % 2 == 0 for n in seq[-1]):
        seq.append(tuple(n // 2 for n in seq[-1]))
    return seq[::-1]


def level_sequence(nc, p, mg_type="h", p_sequence="bisect", keep_duplicate=True):
    """[(cells, degree)] from coarse to fine (element_centered_preconditioners_01.cc:640-672).  The 'ph' / 'hp' lists
    of the reference name the joint level twice (ph: the finest mesh at the lowest degree, hp: the coarsest mesh at
    the highest degree); both copies are levels of the hierarchy, with the identity as transfer between them and a
    smoother each (keep_duplicate=False lists the level once: the hierarchy of rounds 1-2 of this build)."""
    degs = polynomial_coarsening_sequence(p, p_sequence)
    meshes = geometric_coarsening_sequence(nc)
    if mg_type == "h":
        lv = [(m, degs[-1]) for m in meshes]
    elif mg_type == "p":
        lv = [(meshes[-1], q) for q in degs]
    elif mg_type == "hp":
        lv = [(meshes[0], q) for q in degs] + [(m, degs[-1]) for m in meshes]
    elif mg_type == "ph":
        lv = [(m, degs[0]) for m in meshes] + [(meshes[-1], q) for q in degs]
    else:
        raise ValueError(f"Multigrid variant <{mg_type}> is not known!")
    if keep_duplicate:
        return lv
    out = []
    for l in lv:
        if not out or out[-1] != l:
            out.append(l)
    return out


def intermediate_level(seq):
    """the last level of degree 1, else 0 (element_centered_preconditioners_01.cc:677-685): the levels up to it form
    the "intermediate" multigrid (smoothed by "mg intermediate smoother"), which is the coarse-grid solver of the
    multigrid over the levels above (include/multigrid.h:260-375)"""
    il = 0
    for l, (_, deg) in enumerate(seq):
        if deg == 1:
            il = l
    return il


class Multigrid:
    """One V-cycle as a preconditioner (PreconditionMG around deal.II's Multigrid::cycle):
    pre-smoothing = smoother.vmult (zero initial guess), post-smoothing = smoother.step
    (MGSmootherRelaxation with one step), coarse level: coarse.vmult.
    Two nested hierarchies as in include/multigrid.h:260-375: the levels [0, il] (smoothers[1..il] are the
    intermediate smoothers) are one multigrid, applied n_coarse_cycles times as a Richardson iteration
    (SolverRelaxation under IterationNumberControl, multigrid.h:346-365) as the coarse-grid solver of the multigrid
    over the levels [il, L]; with one cycle that is the flat recursion over all levels."""

    def __init__(self, ops, smoothers, coarse, transfers, one_sided=False, il=0, n_coarse_cycles=1):
        self.ops, self.smoothers, self.coarse, self.transfers, self.one_sided = ops, smoothers, coarse, transfers, one_sided
        self.il, self.n_coarse_cycles = il, n_coarse_cycles

    def _cycle(self, level, b, top):
        if level == 0:
            return self.coarse.vmult(b)
        if level == self.il and top and self.n_coarse_cycles > 1:
            # coarse-grid solver of the upper multigrid: n Richardson steps preconditioned by the lower one
            x = np.zeros_like(b)
            for _ in range(self.n_coarse_cycles):
                x = x + self._cycle(level, b - self.ops[level].vmult(x, unit_constrained=True), False)
            return x
        op, sm = self.ops[level], self.smoothers[level]
        x = sm.vmult(b)
        r = b - op.vmult(x, unit_constrained=True)
        bc = self.transfers[level].restrict(r)
        xc = self._cycle(level - 1, bc, True)
        x = x + self.transfers[level].prolongate(xc)
        if not self.one_sided:  # "one-sided v-cycle": identity post-smoother (include/multigrid.h:305-322)
            x = sm.step(x, b)
        return x

    def vmult(self, b, unit_constrained=True):
        return self._cycle(len(self.ops) - 1, b, True)


class CoarseCG:
    """Coarse solve by Jacobi-preconditioned CG to a relative tolerance (stands in for the reference's
    AMG V-cycle, include/precondition.templates.h:592-613; both are approximate coarse inverses)."""

    def __init__(self, op, rel_tol=1e-3, max_iter=200):
        self.op, self.rel_tol, self.max_iter = op, rel_tol, max_iter
        self.jac = orc.JacobiPreconditioner(op)

    def vmult(self, b, unit_constrained=True):
        x, _ = solve_cg(self.op, b, self.jac, rel_tol=self.rel_tol, abs_tol=0.0, max_iter=self.max_iter)
        return x


class CoarseDirect:
    """Coarse solve with the dense level matrix (identity on constrained rows and columns): the twin of
    ASM_MG_COARSE_DIRECT, also in place of the reference's AMG V-cycle."""

    def __init__(self, op):
        n = op.mesh.n_dofs
        A = np.zeros((n, n))
        e = np.zeros(n)
        for j in range(n):
            e[j] = 1.0
            A[:, j] = op.vmult(e, unit_constrained=True)
            e[j] = 0.0
        self.A = A

    def vmult(self, b, unit_constrained=True):
        return np.linalg.solve(self.A, b)


# --------------------------------------------------------------------------
# outer solvers (SolverControl semantics of ReductionControl: stop when the residual norm is
# <= max(abs_tol, rel_tol * initial residual); the count returned is SolverControl::last_step())
# --------------------------------------------------------------------------
def solve_cg(op, b, prec, rel_tol=1e-2, abs_tol=1e-10, max_iter=1000):
    """deal.II SolverCG: preconditioned CG, x_0 = 0, criterion on the unpreconditioned residual"""
    x = np.zeros_like(b)
    r = b.copy()
    r0 = np.linalg.norm(r)
    tol = max(abs_tol, rel_tol * r0)
    if r0 <= tol:
        return x, 0
    z = prec.vmult(r, unit_constrained=True)
    p = z.copy()
    rz = r @ z
    for it in range(1, max_iter + 1):
        Ap = op.vmult(p, unit_constrained=True)
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        if np.linalg.norm(r) <= tol:
            return x, it
        z = prec.vmult(r, unit_constrained=True)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, max_iter


def solve_gmres(op, b, prec, rel_tol=1e-2, abs_tol=1e-10, max_iter=1000, restart=28, right=True):
    """deal.II SolverGMRES with classical Gram-Schmidt; right preconditioning is the reference driver's
    default (element_centered_preconditioners_01.cc:166-168), max_n_tmp_vectors = 30 -> 28 Krylov vectors.
    The residual estimate of the Givens recurrence is the stopping quantity."""
    n = len(b)
    x = np.zeros(n)
    apply_op = lambda v: op.vmult(v, unit_constrained=True)  # noqa: E731
    apply_pc = lambda v: prec.vmult(v, unit_constrained=True)  # noqa: E731
    r = b.copy() if right else apply_pc(b)
    beta = np.linalg.norm(r)
    tol = max(abs_tol, rel_tol * beta)
    it = 0
    if beta <= tol:
        return x, 0
    while it < max_iter:
        V = np.zeros((restart + 1, n))
        H = np.zeros((restart + 1, restart))
        cs, sn, g = np.zeros(restart), np.zeros(restart), np.zeros(restart + 1)
        V[0] = r / beta
        g[0] = beta
        k_done = 0
        for k in range(restart):
            w = apply_op(apply_pc(V[k])) if right else apply_pc(apply_op(V[k]))
            h = V[: k + 1] @ w  # classical Gram-Schmidt
            w = w - h @ V[: k + 1]
            H[: k + 1, k] = h
            H[k + 1, k] = np.linalg.norm(w)
            if H[k + 1, k] > 0:
                V[k + 1] = w / H[k + 1, k]
            for i in range(k):
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            d = np.hypot(H[k, k], H[k + 1, k])
            cs[k], sn[k] = H[k, k] / d, H[k + 1, k] / d
            H[k, k], H[k + 1, k] = d, 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            it += 1
            k_done = k + 1
            if abs(g[k + 1]) <= tol or it >= max_iter:
                break
        y = np.linalg.solve(np.triu(H[:k_done, :k_done]), g[:k_done])
        dx = y @ V[:k_done]
        x = x + (apply_pc(dx) if right else dx)
        if abs(g[k_done]) <= tol or it >= max_iter:
            return x, it
        r = b - apply_op(x)
        if not right:
            r = apply_pc(r)
        beta = np.linalg.norm(r)
    return x, it


class Identity:
    def vmult(self, v, unit_constrained=True):
        return v.copy()

    def is_symmetric(self):
        return True


def right_hand_side_one(mesh: orc.Mesh):
    """LaplaceOperatorMatrixFree::rhs with f = 1 (element_centered_preconditioners_01.cc:65-82, 503) on a
    Cartesian mesh: b_i = int phi_i = product of 1-D integrals; constrained entries are zero"""
    vec = None
    for d in range(mesh.dim):
        h = np.diff(mesh.breaks[d])
        w = np.zeros(mesh.Nu[d])
        wl = mesh.basis.M_ref.sum(axis=1)  # int phi_i on [0, 1]
        for c in range(mesh.nc[d]):
            w[mesh.p * c : mesh.p * c + mesh.p + 1] += h[c] * wl
        vec = w if vec is None else np.multiply.outer(w, vec)  # z, y, x ordering: x fastest
    b = vec.reshape(-1)
    b[mesh.constrained_mask()] = 0.0
    return b


def build_multigrid(nc, p, bc, smoother_factory, coarse_factory, mg_type="h", p_sequence="bisect", mapfn=None, one_sided=False,
                    extent=None, intermediate_factory=None, n_coarse_cycles=1, keep_duplicate=True):
    """levels, operators, transfers and the V-cycle for a box mesh; smoother_factory(level, mesh, op) ("mg smoother"),
    intermediate_factory(level, mesh, op) ("mg intermediate smoother", default: the mg smoother,
    include/precondition.h:140-160) and coarse_factory(mesh, op) return objects with vmult / step"""
    seq = level_sequence(nc, p, mg_type, p_sequence, keep_duplicate)
    il = intermediate_level(seq)
    dim = len(nc)
    meshes = []
    for cells, deg in seq:
        h = None if extent is None else [extent[d] / cells[d] for d in range(dim)]
        meshes.append(orc.Mesh(cells, deg, bc, h=h, mapfn=mapfn))
    ops = [orc.LaplaceOperator(m) for m in meshes]
    transfers = [None] + [Transfer(meshes[l], meshes[l - 1]) for l in range(1, len(meshes))]
    inter = intermediate_factory or smoother_factory
    smoothers = [None] + [(inter if l <= il else smoother_factory)(l, meshes[l], ops[l]) for l in range(1, len(meshes))]
    coarse = coarse_factory(meshes[0], ops[0])
    return Multigrid(ops, smoothers, coarse, transfers, one_sided, il, n_coarse_cycles), meshes, ops
