"""CPU oracle for the matrix-free Laplace operator + FDM additive-Schwarz + Chebyshev
smoother hot path of peterrum/dealii-asm.

*** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.  The product path
(dealii-asm_amd/) never does; it fails loudly if its HIP library is missing.

This is a numpy/scipy RESTATEMENT (not a copy) of the reference algorithm.  The
arithmetic of the reference lives partly in the un-vendored dependency deal.II
(github.com/dealii/dealii, unpinned "master", API level >= 9.5-pre, see
/root/reference/README.md:10); those pieces (FEEvaluation sum factorisation,
TensorProductMatrixSymmetricSum, PreconditionChebyshev, the element-centred
create_laplace_tensor_product_matrix) are restated from their published
mathematics.  Each function cites the reference file:line it follows.

Pinning (see tests/test_oracle_golden.py and DESIGN.md):
  * largest-eigenvalue estimates of P^-1 A printed in
    tests/element_centered_preconitioners/small/*.output (2-D, p=3, 8x8, Dirichlet),
  * the fdm_01.cc invariant (FDM inverse == dense inverse of the restricted
    assembled block on a non-uniform Cartesian mesh),
  * "full-overlap FDM == exact ASM" cross-fixture invariant,
  * subdivided_hyper_cube_balanced_01.output (mesh-size table).
Nothing in the reference pins 3-D p=4 numbers: there parity is transitive
(oracle pinned on the above => HIP kernels checked against the oracle).

Conventions: direction 0 = x is the FASTEST index.  Cell arrays are shaped
(ncells, [k,] j, i).  Unique DoFs are numbered lexicographically
I + Nu_x * (J + Nu_y * K) with Nu_d = nc_d*p (periodic) or nc_d*p+1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg

# --------------------------------------------------------------------------
# 1-D ingredients
# --------------------------------------------------------------------------


def gauss_legendre_01(n):
    """QGauss<1>(n) on [0,1]."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_lobatto_01(n):
    """QGaussLobatto<1>(n) points on [0,1] (support points of FE_Q(n-1))."""
    if n == 2:
        return np.array([0.0, 1.0])
    P = np.polynomial.legendre.Legendre.basis(n - 1)
    xi = np.sort(P.deriv().roots().real)
    # two Newton steps on P'_{n-1} to polish the companion-matrix roots
    dP, ddP = P.deriv(), P.deriv(2)
    for _ in range(2):
        xi = xi - dP(xi) / ddP(xi)
    x = np.concatenate(([-1.0], xi, [1.0]))
    x = 0.5 * (x - x[::-1])  # enforce symmetry
    return 0.5 * (x + 1.0)


def lagrange_eval(nodes, x):
    """V[q,i] = phi_i(x_q), D[q,i] = phi_i'(x_q) for the Lagrange basis on nodes."""
    nodes = np.asarray(nodes, float)
    x = np.atleast_1d(np.asarray(x, float))
    n = len(nodes)
    V = np.ones((len(x), n))
    D = np.zeros((len(x), n))
    for i in range(n):
        denom = 1.0
        for l in range(n):
            if l != i:
                denom *= nodes[i] - nodes[l]
                V[:, i] *= x - nodes[l]
        V[:, i] /= denom
        for m in range(n):
            if m == i:
                continue
            t = np.ones(len(x))
            for l in range(n):
                if l != i and l != m:
                    t *= x - nodes[l]
            D[:, i] += t
        D[:, i] /= denom
    return V, D


@dataclass
class Basis1D:
    """FE_Q<1>(p) on Gauss-Lobatto nodes with QGauss<1>(p+1)
    (element_centered_preconditioners_01.cc:426-427)."""

    p: int
    nodes: np.ndarray = field(init=False)
    xq: np.ndarray = field(init=False)
    wq: np.ndarray = field(init=False)
    N: np.ndarray = field(init=False)  # N[q,i]
    D: np.ndarray = field(init=False)  # D[q,i]
    M_ref: np.ndarray = field(init=False)
    K_ref: np.ndarray = field(init=False)

    def __post_init__(self):
        p = self.p
        self.nodes = gauss_lobatto_01(p + 1)
        self.xq, self.wq = gauss_legendre_01(p + 1)
        self.N, self.D = lagrange_eval(self.nodes, self.xq)
        # deal.II internal::create_reference_mass_and_stiffness_matrices [upstream]
        self.M_ref = np.einsum("qi,qj,q->ij", self.N, self.N, self.wq)
        self.K_ref = np.einsum("qi,qj,q->ij", self.D, self.D, self.wq)


# --------------------------------------------------------------------------
# Geometry maps
# --------------------------------------------------------------------------


def _kershaw_right(eps, x):
    return np.where(x <= 0.5, (2 - eps) * x, 1 + eps * (x - 1))


def _kershaw_left(eps, x):
    return 1 - _kershaw_right(eps, 1 - x)


def _kershaw_step(a, b, x):
    s = a + (b - a) * (x * x * x * (x * (6 * x - 15) + 10))
    return np.where(x <= 0, a, np.where(x >= 1, b, s))


def kershaw(epsy, epsz, pts):
    """Kershaw map on the unit cube (include/kershaw.h:39-80); pts[..., 3]."""
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    layer = np.floor(x * 6.0).astype(int)  # `int layer = x * 6.0`
    lam = (x - layer / 6.0) * 6
    ly, ry = _kershaw_left(epsy, y), _kershaw_right(epsy, y)
    lz, rz = _kershaw_left(epsz, z), _kershaw_right(epsz, z)
    Y = np.where(layer == 0, ly, ry)  # default branch: right
    Z = np.where(layer == 0, lz, rz)
    m14 = (layer == 1) | (layer == 4)
    Y = np.where(m14, _kershaw_step(ly, ry, lam), Y)
    Z = np.where(m14, _kershaw_step(lz, rz, lam), Z)
    m2 = layer == 2
    Y = np.where(m2, _kershaw_step(ry, ly, lam / 2), Y)
    Z = np.where(m2, _kershaw_step(rz, lz, lam / 2), Z)
    m3 = layer == 3
    Y = np.where(m3, _kershaw_step(ry, ly, (1 + lam) / 2), Y)
    Z = np.where(m3, _kershaw_step(rz, lz, (1 + lam) / 2), Z)
    return np.stack([x, Y, Z], axis=-1)


def decompose_for_subdivided_hyper_cube_balanced(dim, s):
    """include/grid_generator.h:107-135 -> (n_refine, subdivisions)."""
    n_refine = s // 6
    rem = s % 6
    sub = [1] * dim
    if rem == 1 and s > 1:
        sub[0], sub[1], sub[2] = 3, 2, 2
        n_refine -= 1
    if rem == 2:
        sub[0] = 2
    elif rem == 3:
        sub[0] = 3
    elif rem == 4:
        sub[0] = sub[1] = 2
    elif rem == 5:
        sub[0], sub[1] = 3, 2
    return n_refine, sub


# --------------------------------------------------------------------------
# Structured mesh + DoF numbering
# --------------------------------------------------------------------------

PERIODIC, DIRICHLET, NEUMANN = "periodic", "dirichlet", "neumann"


class Mesh:
    """Structured hex mesh nc[0] x nc[1] (x nc[2]) with continuous Q_p DoFs.

    geometry: None (Cartesian, cell sizes `h[d]` scalars or per-cell arrays) or
    a callable mapping reference points in cell-index space (x_hat in [0,nc_d])
    to physical points; the geometry is then represented, as in the reference
    (operator.h:592-673 "quadratic geometry", mapping degree 2), by its
    tri-quadratic interpolant on every cell.
    """

    def __init__(self, nc, p, bc, h=None, mapfn=None):
        self.dim = len(nc)
        self.nc = tuple(int(n) for n in nc)
        self.p = int(p)
        self.n1 = p + 1
        bc = [(b, b) if isinstance(b, str) else tuple(b) for b in bc]
        self.bc = bc
        self.periodic = [b[0] == PERIODIC for b in bc]
        for b in bc:
            assert (b[0] == PERIODIC) == (b[1] == PERIODIC)
        self.Nu = tuple(n * p + (0 if per else 1) for n, per in zip(self.nc, self.periodic))
        self.n_dofs = int(np.prod(self.Nu))
        self.ncells = int(np.prod(self.nc))
        self.basis = Basis1D(p)
        # rectilinear breakpoints in cell-index space
        if h is None:
            h = [1.0 / n for n in self.nc]
        self.breaks = []
        for d in range(self.dim):
            hd = np.broadcast_to(np.asarray(h[d], float), (self.nc[d],))
            self.breaks.append(np.concatenate(([0.0], np.cumsum(hd))))
        self.mapfn = mapfn
        self._build_cells()

    # cells are numbered lexicographically, x fastest
    def _build_cells(self):
        dim, nc, p = self.dim, self.nc, self.p
        grids = np.meshgrid(*[np.arange(n) for n in nc[::-1]], indexing="ij")
        # cell coordinate arrays, c[d] flattened with x fastest
        self.cc = [grids[dim - 1 - d].reshape(-1) for d in range(dim)]
        # q2 support points (ncells, [3,]3,3, dim), node axes ordered (k,j,i)
        t = np.array([0.0, 0.5, 1.0])
        pts = np.zeros((self.ncells,) + (3,) * dim + (dim,))
        for d in range(dim):
            shape = [1] * dim
            shape[dim - 1 - d] = 3
            xhat = self.cc[d].reshape((-1,) + (1,) * dim) + t.reshape([1] + shape)
            pts[..., d] = np.broadcast_to(xhat, pts.shape[:-1])
        self.q2_hat = pts
        self.q2_pts = self.to_physical(pts)

    def to_physical(self, xhat):
        out = np.empty_like(xhat)
        for d in range(self.dim):
            out[..., d] = np.interp(xhat[..., d], np.arange(self.nc[d] + 1), self.breaks[d])
        if self.mapfn is not None:
            out = self.mapfn(out)
        return out

    # -- DoF indexing -------------------------------------------------------
    def coord_to_unique(self, d, I):
        """logical 1-D DoF coordinate -> (unique index, exists, dirichlet)."""
        I = np.asarray(I)
        n = self.nc[d] * self.p
        if self.periodic[d]:
            return np.mod(I, n), np.ones(I.shape, bool), np.zeros(I.shape, bool)
        exists = (I >= 0) & (I <= n)
        dirichlet = ((I == 0) & (self.bc[d][0] == DIRICHLET)) | ((I == n) & (self.bc[d][1] == DIRICHLET))
        return np.clip(I, 0, n), exists, dirichlet & exists

    def box_indices(self, origin, m):
        """Global unique indices + validity of the m^dim box with per-cell 1-D
        origins `origin[d]` (arrays of length ncells). valid = exists & !dirichlet."""
        dim = self.dim
        idx = np.zeros((self.ncells,) + (m,) * dim, dtype=np.int64)
        valid = np.ones((self.ncells,) + (m,) * dim, dtype=bool)
        stride = 1
        for d in range(dim):
            I = origin[d][:, None] + np.arange(m)[None, :]
            u, ex, di = self.coord_to_unique(d, I)
            shape = [self.ncells] + [1] * dim
            shape[dim - d] = m
            idx += stride * u.reshape(shape)
            valid &= (ex & ~di).reshape(shape)
            stride *= self.Nu[d]
        return idx, valid

    def cell_dofs(self):
        """(p+1)^dim DoFs of every cell (n_overlap==1 patch)."""
        return self.box_indices([self.p * c for c in self.cc], self.p + 1)

    def constrained_mask(self):
        """unique-DoF mask of homogeneous-Dirichlet-constrained DoFs
        (DoFTools::make_zero_boundary_constraints, operator.h:421-423)."""
        m = np.zeros(self.Nu[::-1], bool)
        for d in range(self.dim):
            ax = self.dim - 1 - d
            if self.bc[d][0] == DIRICHLET:
                sl = [slice(None)] * self.dim
                sl[ax] = 0
                m[tuple(sl)] = True
            if self.bc[d][1] == DIRICHLET:
                sl = [slice(None)] * self.dim
                sl[ax] = -1
                m[tuple(sl)] = True
        return m.reshape(-1)

    def neighbor(self, d, side):
        """cell index of the neighbour in direction d (side -1/+1) or -1."""
        c = self.cc[d] + side
        if self.periodic[d]:
            c = np.mod(c, self.nc[d])
            ok = np.ones_like(c, bool)
        else:
            ok = (c >= 0) & (c < self.nc[d])
        lin = np.zeros(self.ncells, dtype=np.int64)
        stride = 1
        for e in range(self.dim):
            lin += stride * (np.clip(c, 0, self.nc[d] - 1) if e == d else self.cc[e])
            stride *= self.nc[e]
        return np.where(ok, lin, -1)


def apply_1d(mat, T, d, dim):
    """contract `mat[a,b]` with axis of direction d of T (ncells, [k,] j, i)."""
    ax = T.ndim - 1 - d
    out = np.tensordot(T, mat, axes=([ax], [1]))  # contracted axis moves last
    return np.moveaxis(out, -1, ax)


def apply_1d_percell(mats, T, d):
    """like apply_1d with a different matrix per cell: mats (ncells, a, b)."""
    ax = T.ndim - 1 - d
    Tm = np.moveaxis(T, ax, -1)  # (ncells, ..., b)
    out = np.einsum("c...b,cab->c...a", Tm, mats)
    return np.moveaxis(out, -1, ax)


# --------------------------------------------------------------------------
# Laplace operator (matrix-free)
# --------------------------------------------------------------------------


class LaplaceOperator:
    """Restatement of LaplaceOperatorMatrixFree (include/operator.h:266-1628):
    gather -> sum-factorised evaluate(gradients) -> J^-1 J^-T detJ w ->
    integrate(gradients) -> scatter-add (do_cell_integral_global :1335-1351).
    Geometry follows the "quadratic geometry" variant (:592-673, :1035-1159);
    for Cartesian cells it coincides with deal.II's stored Jacobians."""

    def __init__(self, mesh: Mesh):
        self.mesh = mesh
        self.idx, self.valid = mesh.cell_dofs()
        self.constrained = mesh.constrained_mask()
        self._setup_quadratic_geometry()

    @staticmethod
    def q2_monomial_coefficients(mesh):
        """nodal (0,1/2,1) -> monomial coefficients of the tri-quadratic map, direction by direction
        (operator.h:625-658): c0 = x0, c1 = -3x0+4x1-x2, c2 = 2x0-4x1+2x2"""
        dim = mesh.dim
        C = np.array([[1.0, 0.0, 0.0], [-3.0, 4.0, -1.0], [2.0, -4.0, 2.0]])
        v = mesh.q2_pts
        for d in range(dim):
            ax = 1 + (dim - 1 - d)
            v = np.moveaxis(np.tensordot(v, C, axes=([ax], [1])), -1, ax)
        return v  # (ncells, [3,]3,3, dim): X = sum v[k,j,i] x^i y^j z^k

    def _setup_quadratic_geometry(self):
        mesh = self.mesh
        dim = mesh.dim
        v = self.q2_monomial_coefficients(mesh)
        self.q2_coeff = v
        xq, wq = mesh.basis.xq, mesh.basis.wq
        nq = len(xq)
        # Jacobian rows jac[d][e] = dX_e/dxi_d at all quadrature points (:1101-1128)
        P = np.stack([np.ones(nq), xq, xq * xq], 1)  # value  [q, power]
        dP = np.stack([np.zeros(nq), np.ones(nq), 2 * xq], 1)  # derivative
        jac = np.zeros((mesh.ncells,) + (nq,) * dim + (dim, dim))
        for d in range(dim):
            t = self.q2_coeff
            for e in range(dim):
                ax = 1 + (dim - 1 - e)
                t = np.moveaxis(np.tensordot(t, dP if e == d else P, axes=([ax], [1])), -1, ax)
            jac[..., d, :] = t
        det = np.linalg.det(jac)
        self.Ginv = np.linalg.inv(jac)  # do_invert (:877-914)
        w = np.ones((nq,) * dim)
        for d in range(dim):
            shape = [1] * dim
            shape[dim - 1 - d] = nq
            w = w * wq.reshape(shape)
        self.detw = det * w[None]

    def cell_integral(self, U):
        """do_cell_integral_local (:851-864) on U (ncells, [k,] j, i)."""
        mesh = self.mesh
        dim, b = mesh.dim, mesh.basis
        G = []
        for d in range(dim):
            t = U
            for e in range(dim):
                t = apply_1d(b.D if e == d else b.N, t, e, dim)
            G.append(t)
        G = np.stack(G, -1)  # reference gradient at q-points
        tmp = np.einsum("...de,...e->...d", self.Ginv, G) * self.detw[..., None]
        H = np.einsum("...ed,...e->...d", self.Ginv, tmp)
        out = 0
        for d in range(dim):
            t = H[..., d]
            for e in range(dim):
                t = apply_1d((b.D if e == d else b.N).T, t, e, dim)
            out = out + t
        return out

    def vmult(self, src, unit_constrained=False):
        """dst = A src; constrained rows are 0 (plain vmult, operator.h:1353-1365)
        or identity (vmult with a post operation, matrix_free_internal.h:226-255)."""
        U = np.where(self.valid, src[self.idx], 0.0)
        out = self.cell_integral(U)
        dst = np.zeros_like(src)
        np.add.at(dst, self.idx[self.valid], out[self.valid])
        if unit_constrained:
            dst[self.constrained] = src[self.constrained]
        return dst

    def diagonal(self):
        mesh = self.mesh
        n = (mesh.p + 1) ** mesh.dim
        diag = np.zeros(mesh.n_dofs)
        for i in range(n):
            U = np.zeros((mesh.ncells, n))
            U[:, i] = 1.0
            out = self.cell_integral(U.reshape(self.idx.shape)).reshape(mesh.ncells, n)
            sel = self.valid.reshape(mesh.ncells, n)[:, i]
            np.add.at(diag, self.idx.reshape(mesh.ncells, n)[sel, i], out[sel, i])
        return diag

    def inverse_diagonal(self):
        """compute_inverse_diagonal (operator.h:1512-1524)."""
        d = self.diagonal()
        return np.where(np.abs(d) > 1.0e-10, 1.0 / np.where(d == 0, 1, d), 1.0)

    def dense(self):
        n = self.mesh.n_dofs
        A = np.zeros((n, n))
        e = np.zeros(n)
        for i in range(n):
            e[i] = 1.0
            A[:, i] = self.vmult(e)
            e[i] = 0.0
        return A


# --------------------------------------------------------------------------
# FDM additive Schwarz preconditioner
# --------------------------------------------------------------------------


def harmonic_cell_extent(mesh: Mesh):
    """GridTools::compute_harmonic_cell_extend (grid_tools.h:11-50): integral over
    the (p+1)^(dim-1) Gauss face quadrature of the distance between matching
    points on faces 2d and 2d+1, using the (Q2) mapping."""
    dim = mesh.dim
    xq, wq = mesh.basis.xq, mesh.basis.wq
    L, _ = lagrange_eval([0.0, 0.5, 1.0], xq)  # [q, node]
    L0, _ = lagrange_eval([0.0, 0.5, 1.0], [0.0, 1.0])  # [face, node]
    ext = np.zeros((mesh.ncells, dim))
    for d in range(dim):
        t = mesh.q2_pts
        for e in range(dim):
            ax = 1 + (dim - 1 - e)
            t = np.moveaxis(np.tensordot(t, L0 if e == d else L, axes=([ax], [1])), -1, ax)
        axd = 1 + (dim - 1 - d)
        diff = np.take(t, 1, axis=axd) - np.take(t, 0, axis=axd)
        dist = np.sqrt((diff * diff).sum(-1))  # (ncells, [q,] q)
        w = np.ones(())
        for _ in range(dim - 1):
            w = np.multiply.outer(w, wq)
        ext[:, d] = (dist * w[None]).reshape(mesh.ncells, -1).sum(1)
    return ext


def harmonic_patch_extent(mesh: Mesh):
    """compute_harmonic_patch_extend (grid_tools.h:54-138): [cell][d][left,self,right];
    0 where there is no neighbour."""
    ext = harmonic_cell_extent(mesh)
    out = np.zeros((mesh.ncells, mesh.dim, 3))
    for d in range(mesh.dim):
        out[:, d, 1] = ext[:, d]
        for s, side in ((0, -1), (2, +1)):
            nb = mesh.neighbor(d, side)
            out[:, d, s] = np.where(nb >= 0, ext[np.maximum(nb, 0), d], 0.0)
    return out


def fdm_matrices_element(b: Basis1D, h3, bc_lo, bc_hi, n_overlap):
    """Element-centred 1-D mass/stiffness with overlap [deal.II upstream
    TensorProductMatrixCreator::create_laplace_tensor_product_matrix, call site
    matrix_free.h:350-360; pinned by the fdm_01.cc invariant]: principal
    sub-matrix of the assembled 1-D matrices on (left, self, right) restricted to
    the m = p-1+2n patch DoFs; Dirichlet face DoF row/col cleared; rows of
    non-existing DoFs stay zero."""
    n1 = b.p + 1
    m = n1 - 2 + 2 * n_overlap
    M = np.zeros((m, m))
    K = np.zeros((m, m))
    o = n_overlap - 1
    M[o : o + n1, o : o + n1] = b.M_ref * h3[1]
    K[o : o + n1, o : o + n1] = b.K_ref / h3[1]
    if bc_lo == "internal":
        M[:n_overlap, :n_overlap] += b.M_ref[n1 - n_overlap :, n1 - n_overlap :] * h3[0]
        K[:n_overlap, :n_overlap] += b.K_ref[n1 - n_overlap :, n1 - n_overlap :] / h3[0]
    elif bc_lo == DIRICHLET:
        M[o, :] = M[:, o] = 0.0
        K[o, :] = K[:, o] = 0.0
    if bc_hi == "internal":
        s = n_overlap + n1 - 2
        M[s:, s:] += b.M_ref[:n_overlap, :n_overlap] * h3[2]
        K[s:, s:] += b.K_ref[:n_overlap, :n_overlap] / h3[2]
    elif bc_hi == DIRICHLET:
        s = n_overlap + n1 - 2
        M[s, :] = M[:, s] = 0.0
        K[s, :] = K[:, s] = 0.0
    return M, K


def fdm_matrices_vertex(b: Basis1D, h2):
    """Vertex-star 1-D matrices (include/tensor_product_matrix_creator.h:7-61)."""
    n1 = b.p + 1
    m = 2 * (n1 - 1) - 1
    M = np.zeros((m, m))
    K = np.zeros((m, m))
    M[: n1 - 1, : n1 - 1] += b.M_ref[1:, 1:] * h2[0]
    K[: n1 - 1, : n1 - 1] += b.K_ref[1:, 1:] / h2[0]
    M[n1 - 2 :, n1 - 2 :] += b.M_ref[:-1, :-1] * h2[1]
    K[n1 - 2 :, n1 - 2 :] += b.K_ref[:-1, :-1] / h2[1]
    return M, K


def fdm_eigen(M, K):
    """Generalised eigen-decomposition K q = lambda M q, Q^T M Q = I, with rows
    of constrained (zero-diagonal) DoFs treated as deal.II's
    TensorProductMatrixSymmetricSum does [upstream spectral_assembly]: unit
    diagonal during the solve, eigenvector rows zeroed afterwards."""
    m = M.shape[0]
    con = np.diag(M) == 0.0
    Mc, Kc = M.copy(), K.copy()
    Mc[con, con] = 1.0
    Kc[con, con] = 1.0
    lam, Q = scipy.linalg.eigh(Kc, Mc)
    Q[con, :] = 0.0
    return lam, Q  # Q[i, a]: component i of eigenvector a


def fdm_eigen_batched(M, K):
    """fdm_eigen for a stack of matrix pairs (..., m, m) at once (meshes whose cells all differ - Kershaw - have one
    pair per cell and direction; a Python loop over them does not reach the sizes of the benchmarks): Cholesky
    M = L L^T, symmetric eigen-problem of L^-1 K L^-T, Q = L^-T V, so that Q^T M Q = I as scipy.linalg.eigh(K, M)
    normalises.  Signs of the eigenvectors may differ from fdm_eigen's; Q diag(.) Q^T does not depend on them."""
    m = M.shape[-1]
    con = np.einsum("...ii->...i", M) == 0.0
    Mc, Kc = M.copy(), K.copy()
    ii = np.arange(m)
    Mc[..., ii, ii] = np.where(con, 1.0, Mc[..., ii, ii])
    Kc[..., ii, ii] = np.where(con, 1.0, Kc[..., ii, ii])
    L = np.linalg.cholesky(Mc)
    Li = np.linalg.inv(L)
    A = Li @ Kc @ np.swapaxes(Li, -1, -2)
    A = 0.5 * (A + np.swapaxes(A, -1, -2))
    lam, V = np.linalg.eigh(A)
    Q = np.swapaxes(Li, -1, -2) @ V
    Q = np.where(con[..., :, None], 0.0, Q)
    return lam, Q


def fdm_matrices_element_batched(b: "Basis1D", h3, lo_internal, lo_dirichlet, hi_internal, hi_dirichlet, n_overlap):
    """fdm_matrices_element for arrays: h3 (..., 3), boolean arrays for the kind of the two faces"""
    n1 = b.p + 1
    m = n1 - 2 + 2 * n_overlap
    shape = h3.shape[:-1]
    M = np.zeros(shape + (m, m))
    K = np.zeros(shape + (m, m))
    o = n_overlap - 1
    hs = h3[..., 1][..., None, None]
    M[..., o : o + n1, o : o + n1] = b.M_ref * hs
    K[..., o : o + n1, o : o + n1] = b.K_ref / hs
    hl = np.where(lo_internal, h3[..., 0], 1.0)[..., None, None]
    li = lo_internal[..., None, None]
    M[..., :n_overlap, :n_overlap] += np.where(li, b.M_ref[n1 - n_overlap :, n1 - n_overlap :] * hl, 0.0)
    K[..., :n_overlap, :n_overlap] += np.where(li, b.K_ref[n1 - n_overlap :, n1 - n_overlap :] / hl, 0.0)
    keep = np.ones(shape + (m,))
    keep[..., o] = np.where(lo_dirichlet, 0.0, 1.0)
    s = n_overlap + n1 - 2
    hr = np.where(hi_internal, h3[..., 2], 1.0)[..., None, None]
    hi = hi_internal[..., None, None]
    M[..., s:, s:] += np.where(hi, b.M_ref[:n_overlap, :n_overlap] * hr, 0.0)
    K[..., s:, s:] += np.where(hi, b.K_ref[:n_overlap, :n_overlap] / hr, 0.0)
    keep[..., s] = np.where(hi_dirichlet, 0.0, keep[..., s])
    mask = keep[..., :, None] * keep[..., None, :]
    return M * mask, K * mask


class ASPoissonPreconditioner:
    """Restatement of ASPoissonPreconditioner (include/matrix_free.h:63-1568).

    element_centric: patch = cell + (n_overlap-1) DoF layers (m = p-1+2n,
    :90); else vertex-star patch anchored at the cell's upper vertex
    (m = 2p-1, :91; dof_tools.h:206-300).  weight_type in
    none|pre|post|symm|ras (restrictors.h:8-15)."""

    def __init__(self, mesh: Mesh, n_overlap=1, weight_type="post", element_centric=True, batched_eigen=None):
        """batched_eigen: the eigen-decompositions of all cells at once (fdm_eigen_batched) instead of a loop with a cache
        of distinct matrix pairs; default: on for element-centred patches of more than 4096 cells"""
        self.mesh = mesh
        p, dim = mesh.p, mesh.dim
        self.n_overlap = n_overlap = min(n_overlap, p)  # precondition.templates.h:195-196
        self.weight_type = weight_type
        self.element_centric = element_centric
        self.m = m = (p + 2 * n_overlap - 1) if element_centric else (2 * p - 1)
        b = mesh.basis
        if element_centric:
            origin = [p * c - (n_overlap - 1) for c in mesh.cc]
        else:
            origin = [p * c + 1 for c in mesh.cc]
        self.idx, self.valid = mesh.box_indices(origin, m)
        if not element_centric:
            # all-invalid if any of the 2^dim cells is missing (dof_tools.h:223-227)
            ok = np.ones(mesh.ncells, bool)
            for d in range(dim):
                ok &= mesh.neighbor(d, +1) >= 0
            self.valid &= ok.reshape((-1,) + (1,) * dim)
        self.constrained = mesh.constrained_mask()

        ext = harmonic_patch_extent(mesh)  # matrix_free.h:220-223
        lam = np.zeros((mesh.ncells, dim, m))
        Q = np.zeros((mesh.ncells, dim, m, m))
        cache = {}
        if batched_eigen is None:
            batched_eigen = element_centric and mesh.ncells > 4096
        if batched_eigen:
            assert element_centric
            for d in range(dim):
                lo_n, hi_n = mesh.neighbor(d, -1) >= 0, mesh.neighbor(d, +1) >= 0
                Md, Kd = fdm_matrices_element_batched(b, ext[:, d], lo_n, ~lo_n & (mesh.bc[d][0] == DIRICHLET), hi_n,
                                                      ~hi_n & (mesh.bc[d][1] == DIRICHLET), n_overlap)
                lam[:, d], Q[:, d] = fdm_eigen_batched(Md, Kd)
        for c in range(0 if batched_eigen else mesh.ncells):
            for d in range(dim):
                if element_centric:
                    lo = "internal" if mesh.neighbor(d, -1)[c] >= 0 else mesh.bc[d][0]
                    hi = "internal" if mesh.neighbor(d, +1)[c] >= 0 else mesh.bc[d][1]
                    key = (tuple(ext[c, d]), lo, hi)
                else:
                    # collect_patch_extend (matrix_free.h:1510-1524)
                    h2 = [ext[c, d, 1] if ext[c, d, 1] != 0 else 1.0, ext[c, d, 2] if ext[c, d, 2] != 0 else 1.0]
                    key = (tuple(h2),)
                if key not in cache:
                    if element_centric:
                        Md, Kd = fdm_matrices_element(b, ext[c, d], lo, hi, n_overlap)
                    else:
                        Md, Kd = fdm_matrices_vertex(b, h2)
                    cache[key] = fdm_eigen(Md, Kd)
                lam[c, d], Q[c, d] = cache[key]
        self.lam, self.Q = lam, Q
        self.n_instances_1d = len(cache) if not batched_eigen else None
        self._setup_weights()

    def _setup_weights(self):
        """matrix_free.h:536-712."""
        mesh, wt = self.mesh, self.weight_type
        self.w_pre = self.w_post = None
        if wt == "none":
            return
        if wt == "ras":
            # owner = smallest cell id among patches containing the DoF in their
            # core region (:575-608); cells are numbered lexicographically here.
            m, p, n = self.m, mesh.p, self.n_overlap
            core = np.ones((m,) * mesh.dim, bool)
            if self.element_centric:
                r = np.arange(m)
                c1 = (r >= n - 1) & (r < p + n)
                for d in range(mesh.dim):
                    shape = [1] * mesh.dim
                    shape[mesh.dim - 1 - d] = m
                    core = core & c1.reshape(shape)
            owner = np.full(mesh.n_dofs, mesh.ncells + 1, dtype=np.int64)
            ids = np.broadcast_to(np.arange(mesh.ncells).reshape((-1,) + (1,) * mesh.dim), self.idx.shape)
            sel = self.valid & core[None]
            np.minimum.at(owner, self.idx[sel], ids[sel])
            self.w_post = (owner[self.idx] == ids) & self.valid
            self.w_post = self.w_post.astype(float)
            return
        val = np.zeros(mesh.n_dofs)
        np.add.at(val, self.idx[self.valid], 1.0)  # :676-698
        if wt == "symm":
            w = np.where(val == 0, 0.0, 1.0 / np.sqrt(np.where(val == 0, 1, val)))
        else:
            w = np.where(val == 0, 0.0, 1.0 / np.where(val == 0, 1, val))
        self.weights = w
        wl = np.where(self.valid, w[self.idx], 0.0)
        if wt in ("pre", "symm"):
            self.w_pre = wl
        if wt in ("post", "symm"):
            self.w_post = wl

    def apply_inverse(self, T):
        """TensorProductMatrixSymmetricSum::apply_inverse [upstream]:
        (Q..Q) diag(1/(lam_i+lam_j+lam_k)) (Q..Q)^T."""
        dim = self.mesh.dim
        for d in range(dim):
            T = apply_1d_percell(np.swapaxes(self.Q[:, d], 1, 2), T, d)
        s = 0
        for d in range(dim):
            shape = [self.mesh.ncells] + [1] * dim
            shape[dim - d] = self.m
            s = s + self.lam[:, d].reshape(shape)
        T = T / s
        for d in range(dim):
            T = apply_1d_percell(self.Q[:, d], T, d)
        return T

    def vmult(self, src, unit_constrained=False):
        """dst = sum_c R_c^T W_post A_c^-1 W_pre R_c src (vmult_internal :1007-1364)."""
        T = np.where(self.valid, src[self.idx], 0.0)
        if self.w_pre is not None:
            T = T * self.w_pre
        T = self.apply_inverse(T)
        if self.w_post is not None:
            T = T * self.w_post
        dst = np.zeros_like(src)
        np.add.at(dst, self.idx[self.valid], T[self.valid])
        if unit_constrained:
            dst[self.constrained] = src[self.constrained]
        return dst

    def is_symmetric(self):
        return self.weight_type in ("none", "symm")  # :896-904

    def dense(self):
        n = self.mesh.n_dofs
        A = np.zeros((n, n))
        e = np.zeros(n)
        for i in range(n):
            e[i] = 1.0
            A[:, i] = self.vmult(e)
            e[i] = 0.0
        return A


class JacobiPreconditioner:
    """DiagonalMatrix / DiagonalMatrixPrePost (preconditioners.h:951-997) around
    compute_inverse_diagonal (operator.h:1512-1524)."""

    def __init__(self, op: LaplaceOperator):
        self.inv_diag = op.inverse_diagonal()

    def vmult(self, src, unit_constrained=False):
        return src * self.inv_diag

    def is_symmetric(self):
        return True


# --------------------------------------------------------------------------
# Chebyshev smoother (deal.II PreconditionChebyshev, configured at
# precondition.templates.h:89-158) [upstream, restated]
# --------------------------------------------------------------------------


def initial_guess(n, constrained):
    """deal.II set_initial_guess for distributed vectors: (i % 11) minus mean,
    then constraints.set_zero."""
    v = (np.arange(n) % 11).astype(float)
    v -= v.mean()
    v[constrained] = 0.0
    return v


def power_iteration(op, prec, v, n_iterations):
    """deal.II internal::PreconditionChebyshevImplementation::power_iteration."""
    v = v / np.linalg.norm(v)
    est = 0.0
    for _ in range(n_iterations):
        t = prec.vmult(op.vmult(v))
        est = float(v @ t)
        v = t / np.linalg.norm(t)
    return est


def lanczos_max_eigenvalue(op, prec, b, n_iterations, tol=1e-7):
    """Largest Ritz value of P^-1 A from <= n_iterations preconditioned CG steps
    started at x=0 with rhs b (deal.II SolverCG eigenvalue slot, eig_cg_residual=1e-7)."""
    x = np.zeros_like(b)
    r = b.copy()
    z = prec.vmult(r)
    pvec = z.copy()
    rz = r @ z
    r0 = np.linalg.norm(r)
    alphas, betas = [], []
    for _ in range(n_iterations):
        Ap = op.vmult(pvec)
        alpha = rz / (pvec @ Ap)
        x += alpha * pvec
        r -= alpha * Ap
        alphas.append(alpha)
        if np.linalg.norm(r) < tol * r0:
            break
        z = prec.vmult(r)
        rz_new = r @ z
        beta = rz_new / rz
        betas.append(beta)
        rz = rz_new
        pvec = z + beta * pvec
    k = len(alphas)
    T = np.zeros((k, k))
    for i in range(k):
        T[i, i] = 1.0 / alphas[i] + (betas[i - 1] / alphas[i - 1] if i > 0 else 0.0)
        if i + 1 < k:
            T[i, i + 1] = T[i + 1, i] = math.sqrt(betas[i]) / alphas[i]
    ev = np.linalg.eigvalsh(T)
    return float(ev[-1]), float(ev[0])


class Chebyshev:
    """Chebyshev smoother; vmult == zero initial guess, step == given initial guess.
    The operator and preconditioner act as identity on constrained DoFs, which is what the
    reference's optimize=2 path does through the pre/post overloads.
    polynomial_type "1st kind" | "4th kind" (precondition.templates.h:135-152).  The 4th-kind
    recurrence [deal.II PreconditionChebyshev, PolynomialType::fourth_kind; Lottes, "Optimal
    polynomial smoothers for multigrid V-cycles", 2022, Algorithm 1] only uses the upper bound
    lambda_max = 1.2 * estimate:
        x_1     = x_0 + 4/(3 lambda_max) P^-1 r_0
        x_{j+1} = x_j + (2j-1)/(2j+3) (x_j - x_{j-1}) + (8j+4)/(2j+3)/lambda_max P^-1 r_j
    (tests/test_oracle_golden.py pins it to the closed form of the 4th-kind polynomial)."""

    def __init__(self, op, prec, degree=3, smoothing_range=20.0, eig_n_iterations=40, ev_algorithm=None, lambda_max=None,
                 polynomial_type="1st kind"):
        self.op, self.prec, self.degree = op, prec, degree
        assert polynomial_type in ("1st kind", "4th kind")
        self.fourth = polynomial_type == "4th kind"
        con = op.constrained
        if lambda_max is None:
            v = initial_guess(op.mesh.n_dofs, con)
            if ev_algorithm is None:
                ev_algorithm = "lanczos" if prec.is_symmetric() else "power iteration"
            if ev_algorithm == "lanczos":
                lambda_max, _ = lanczos_max_eigenvalue(op, prec, v, eig_n_iterations)
            else:
                lambda_max = power_iteration(op, prec, v, eig_n_iterations)
        self.lambda_hat = lambda_max
        self.max_ev = 1.2 * lambda_max  # safety factor
        self.min_ev = self.max_ev / smoothing_range
        self.theta = 0.5 * (self.max_ev + self.min_ev)
        self.delta = self.max_ev if self.fourth else 0.5 * (self.max_ev - self.min_ev)

    def _run(self, x, b, zero_start):
        op, prec = self.op, self.prec
        theta, delta = self.theta, self.delta
        first = 4.0 / (3.0 * delta) if self.fourth else 1.0 / theta
        if zero_start:
            x = first * prec.vmult(b, unit_constrained=True)
        else:
            r = b - op.vmult(x, unit_constrained=True)
            x = x + first * prec.vmult(r, unit_constrained=True)
        if self.degree < 2:
            return x
        rhok, sigma = delta / theta, theta / delta
        x_prev = np.zeros_like(x) if zero_start else self._x0
        for j in range(1, self.degree):
            if self.fourth:
                f1 = (2.0 * j - 1.0) / (2.0 * j + 3.0)
                f2 = (8.0 * j + 4.0) / (delta * (2.0 * j + 3.0))
            else:
                rhokp = 1.0 / (2.0 * sigma - rhok)
                f1 = rhokp * rhok
                f2 = 2.0 * rhokp / delta
                rhok = rhokp
            r = b - op.vmult(x, unit_constrained=True)
            t = prec.vmult(r, unit_constrained=True)
            x_new = x + f1 * (x - x_prev) + f2 * t
            x_prev, x = x, x_new
        return x

    def vmult(self, b, unit_constrained=True):
        return self._run(None, b, True)

    def step(self, x, b):
        self._x0 = x.copy()
        return self._run(x.copy(), b, False)
