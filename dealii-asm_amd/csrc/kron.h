// Host side of the wave-pair sweep (kernels_pair.hip): the tables of the Kronecker-diagonal cell operator (grid.h:
// DevKron) for the FDM preconditioner (TensorProductMatrixSymmetricSum::apply_inverse with the weights of
// ASPoissonPreconditioner, include/matrix_free.h:1023-1062, folded into the 1-D matrices) and for the Laplace operator on
// box cells (include/operator.h:866-875 as the Kronecker sum of the 1-D reference matrices, kernels_tiled.hip).
#pragma once

#include "numerics.h"
#include "setup.h"

#include <array>
#include <cmath>
#include <map>
#include <vector>

namespace asmamd
{
  struct HostKron
  {
    bool                valid = false;
    bool                mask_constrained = false; // the sweep reads constrained DoFs of the source as zero
    int                 n_cls[3] = {0, 0, 0};
    std::vector<int>    cls[3]; // [nc + 2]
    std::vector<int>    sym;    // [3][KRON_MAX_CLS]
    std::vector<double> mat;    // [3][KRON_MAX_CLS][KRON_MAT_STRIDE]
    std::vector<double> diag;   // [clz][cly][clx][KRON_DIAG_STRIDE]
  };

  namespace kron_detail
  {
    // F[i][a], B[i][a] -> the block of one (direction, class): full matrices and their even-odd forms
    inline void
    store_matrices(HostKron &k, const int d, const int c, const double F[25], const double B[25], const bool sym)
    {
      double *m = &k.mat[(size_t)(d * KRON_MAX_CLS + c) * KRON_MAT_STRIDE];
      // full forms, one output per row of six (16-byte aligned rows): forward out[a] = sum_i F[i][a] in[i] reads row a of
      // F^T, backward out[i] = sum_a B[i][a] in[a] reads row i of B
      for (int r = 0; r < 5; ++r)
        for (int q = 0; q < 5; ++q)
          {
            m[r * 6 + q]      = F[q * 5 + r];
            m[30 + r * 6 + q] = B[r * 5 + q];
          }
      // even-odd forms: E[a][i] = q[i][a] (a, i < 3), O[a][i] = q[i][3 + a] (a, i < 2)
      for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 3; ++i)
          {
            m[60 + a * 3 + i] = F[i * 5 + a];
            m[76 + a * 3 + i] = B[i * 5 + a];
          }
      for (int a = 0; a < 2; ++a)
        for (int i = 0; i < 2; ++i)
          {
            m[60 + 9 + a * 2 + i] = F[i * 5 + 3 + a];
            m[76 + 9 + a * 2 + i] = B[i * 5 + 3 + a];
          }
      k.sym[d * KRON_MAX_CLS + c] = sym ? 1 : 0;
    }
    // columns 0..2 even, 3..4 odd vectors, bitwise
    inline bool
    parity_sorted(const double Q[25])
    {
      for (int a = 0; a < 5; ++a)
        for (int i = 0; i < 5; ++i)
          if (Q[i * 5 + a] != (a < 3 ? 1. : -1.) * Q[(4 - i) * 5 + a])
            return false;
      return true;
    }
    inline void
    init(HostKron &k)
    {
      k.valid            = false;
      k.mask_constrained = false;
      k.sym.assign(3 * KRON_MAX_CLS, 0);
      k.mat.assign((size_t)3 * KRON_MAX_CLS * KRON_MAT_STRIDE, 0.);
      k.diag.assign((size_t)KRON_MAX_CLS * KRON_MAX_CLS * KRON_MAX_CLS * KRON_DIAG_STRIDE, 0.);
    }
  } // namespace kron_detail

  // FDM preconditioner, degree 4, cell patches with n_overlap 1 on a tensor-product mesh
  inline bool
  build_kron_fdm(const HostSetup &s, HostKron &k)
  {
    using namespace kron_detail;
    init(k);
    if (s.g.p != 4 || s.m != 5 || !s.tensor_inst || s.prm.element_centric == 0 || s.prm.precond_type != ASM_PRECOND_FDM)
      return false;
    const int           E = 30;
    std::vector<double> lam[3]; // eigenvalues per class
    for (int d = 0; d < 3; ++d)
      {
        std::map<std::array<double, 11>, int> ids;
        k.cls[d].assign(s.ence[d], 0);
        for (int c = -1; c <= s.g.nc[d]; ++c)
          {
            if (s.h[d][c + 1] == 0.) // the cell does not exist
              continue;
            const int             inst = s.inst[d][c + 1];
            std::array<double, 11> key;
            key[0] = inst;
            for (int i = 0; i < 5; ++i)
              {
                key[1 + i] = s.has_pre ? s.w_pre[d][(size_t)(c + 1) * 5 + i] : 1.;
                key[6 + i] = s.has_post ? s.w_post[d][(size_t)(c + 1) * 5 + i] : 1.;
              }
            const auto it = ids.emplace(key, (int)ids.size());
            if (it.second)
              {
                const int id = it.first->second;
                if (id >= KRON_MAX_CLS)
                  return false;
                const double *Q = &s.eig[(size_t)inst * E];
                double        F[25], B[25];
                bool          wsym = true;
                for (int i = 0; i < 5; ++i)
                  {
                    wsym = wsym && key[1 + i] == key[1 + 4 - i] && key[6 + i] == key[6 + 4 - i];
                    for (int a = 0; a < 5; ++a)
                      {
                        F[i * 5 + a] = key[1 + i] * Q[i * 5 + a];
                        B[i * 5 + a] = key[6 + i] * Q[i * 5 + a];
                      }
                  }
                store_matrices(k, d, id, F, B, s.eig_sym[inst] != 0 && wsym && parity_sorted(Q));
                lam[d].resize((size_t)(id + 1) * 5);
                for (int a = 0; a < 5; ++a)
                  lam[d][(size_t)id * 5 + a] = Q[25 + a];
              }
            k.cls[d][c + 1] = it.first->second;
          }
        k.n_cls[d] = (int)ids.size();
      }
    for (int cz = 0; cz < k.n_cls[2]; ++cz)
      for (int cy = 0; cy < k.n_cls[1]; ++cy)
        for (int cx = 0; cx < k.n_cls[0]; ++cx)
          {
            double *D = &k.diag[(size_t)((cz * KRON_MAX_CLS + cy) * KRON_MAX_CLS + cx) * KRON_DIAG_STRIDE];
            for (int c = 0; c < 5; ++c)
              for (int b = 0; b < 5; ++b)
                for (int a = 0; a < 5; ++a)
                  D[(b * 5 + a) * 5 + c] = 1.0 / (lam[0][(size_t)cx * 5 + a] + (lam[1][(size_t)cy * 5 + b] + lam[2][(size_t)cz * 5 + c]));
          }
    k.valid = true;
    return true;
  }

  // Laplace operator, degree 4, box cells of one size per direction: ONE class per direction (F = B = F_ref, even / odd
  // columns); the sweep zeroes the constrained DoFs of the source itself (DevKron::mask_constrained), so that cells at a
  // Dirichlet face need no matrices of their own - and no sweeps with full matrices
  inline bool
  build_kron_laplace(const HostSetup &s, const Basis1D &b, HostKron &k)
  {
    using namespace kron_detail;
    init(k);
    if (s.g.p != 4 || !s.cartesian || b.F_ref.size() != 25 || b.lam_ref.size() != 5 || !parity_sorted(b.F_ref.data()))
      return false;
    double h[3] = {0., 0., 0.};
    for (int d = 0; d < 3; ++d)
      {
        k.cls[d].assign(s.ence[d], 0);
        for (int c = -1; c <= s.g.nc[d]; ++c)
          {
            if (s.h[d][c + 1] == 0.)
              continue;
            if (h[d] == 0.)
              h[d] = s.h[d][c + 1];
            else if (h[d] != s.h[d][c + 1])
              return false; // graded mesh: several diagonals per direction (tiled kernel)
          }
        store_matrices(k, d, 0, b.F_ref.data(), b.F_ref.data(), true);
        k.n_cls[d] = 1;
      }
    // hy hz / hx la + hx hz / hy lb + hx hy / hz lc (kernels_tiled.hip: laplace_cart_tiled_kernel)
    const double alpha = h[1] / h[0] * h[2], beta = h[0] / h[1] * h[2], gamma = h[0] * h[1] / h[2];
    double *     D = &k.diag[0];
    for (int c = 0; c < 5; ++c)
      for (int bb = 0; bb < 5; ++bb)
        for (int a = 0; a < 5; ++a)
          D[(bb * 5 + a) * 5 + c] = std::fma(alpha, b.lam_ref[a], std::fma(beta, b.lam_ref[bb], gamma * b.lam_ref[c]));
    k.mask_constrained = true;
    k.valid            = true;
    return true;
  }
} // namespace asmamd
