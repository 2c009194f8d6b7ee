// Host-side 1-D numerics: quadrature, Lagrange bases, reference mass/stiffness
// matrices and the generalized symmetric eigen-solver used by the FDM setup.
//
// Reference call sites these replace (the implementations live in deal.II):
//   FE_Q<1>(p) + QGauss<1>(p+1)      element_centered_preconditioners_01.cc:426-427
//   internal::create_reference_mass_and_stiffness_matrices
//                                    include/tensor_product_matrix_creator.h:16-24
//   TensorProductMatrixSymmetricSumCollection::insert/finalize (LAPACK sygv)
//                                    include/matrix_free.h:389-392
#pragma once

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <vector>

namespace asmamd
{
  // QGauss<1>(n) on [0,1]
  inline void
  gauss_legendre(const int n, std::vector<double> &x, std::vector<double> &w)
  {
    x.assign(n, 0.);
    w.assign(n, 0.);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < n; ++i)
      {
        long double z = std::cos(pi * (i + 0.75L) / (n + 0.5L));
        long double pp = 0;
        for (int it = 0; it < 100; ++it)
          {
            long double p1 = 1, p2 = 0;
            for (int j = 0; j < n; ++j)
              {
                const long double p3 = p2;
                p2                   = p1;
                p1 = ((2 * j + 1) * z * p2 - j * p3) / (j + 1);
              }
            pp                  = n * (z * p1 - p2) / (z * z - 1);
            const long double dz = p1 / pp;
            z -= dz;
            if (std::fabs((double)dz) < 1e-19)
              break;
          }
        // ascending order on [0,1]
        x[n - 1 - i] = (double)(0.5L * (1 + z));
        w[n - 1 - i] = (double)(1.0L / ((1 - z * z) * pp * pp));
      }
    for (int i = 0; i < n / 2; ++i) // enforce symmetry
      {
        const double a = 0.5 * (x[i] + (1. - x[n - 1 - i]));
        x[i]           = a;
        x[n - 1 - i]   = 1. - a;
        const double b = 0.5 * (w[i] + w[n - 1 - i]);
        w[i] = w[n - 1 - i] = b;
      }
    if (n % 2)
      x[n / 2] = 0.5;
  }

  // QGaussLobatto<1>(n) points on [0,1] (support points of FE_Q(n-1))
  inline std::vector<double>
  gauss_lobatto(const int n)
  {
    std::vector<double> x(n);
    x[0]     = 0.;
    x[n - 1] = 1.;
    const int          N  = n - 1; // roots of P_N'
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 1; i < n - 1; ++i)
      {
        long double z = -std::cos(pi * i / N);
        for (int it = 0; it < 100; ++it)
          {
            // P_N, P_{N-1} by recurrence
            long double p1 = 1, p2 = 0;
            for (int j = 0; j < N; ++j)
              {
                const long double p3 = p2;
                p2                   = p1;
                p1 = ((2 * j + 1) * z * p2 - j * p3) / (j + 1);
              }
            // f = P_N' = N (z P_N - P_{N-1}) / (z^2-1);
            // (1-z^2) P_N'' = 2 z P_N' - N(N+1) P_N
            const long double dp  = N * (z * p1 - p2) / (z * z - 1);
            const long double ddp = (2 * z * dp - N * (N + 1) * p1) / (1 - z * z);
            const long double dz  = dp / ddp;
            z -= dz;
            if (std::fabs((double)dz) < 1e-19)
              break;
          }
        x[i] = (double)(0.5L * (1 + z));
      }
    for (int i = 0; i < n / 2; ++i)
      {
        const double a = 0.5 * (x[i] + (1. - x[n - 1 - i]));
        x[i]           = a;
        x[n - 1 - i]   = 1. - a;
      }
    if (n % 2)
      x[n / 2] = 0.5;
    return x;
  }

  // V[q*n+i] = phi_i(x_q), D[q*n+i] = phi_i'(x_q), Lagrange basis on `nodes`
  inline void
  lagrange_eval(const std::vector<double> &nodes,
                const std::vector<double> &x,
                std::vector<double> &      V,
                std::vector<double> &      D)
  {
    const int n = nodes.size(), nq = x.size();
    V.assign(nq * n, 0.);
    D.assign(nq * n, 0.);
    for (int q = 0; q < nq; ++q)
      for (int i = 0; i < n; ++i)
        {
          long double denom = 1, v = 1, d = 0;
          for (int l = 0; l < n; ++l)
            if (l != i)
              {
                denom *= (long double)nodes[i] - nodes[l];
                v *= (long double)x[q] - nodes[l];
              }
          for (int m = 0; m < n; ++m)
            {
              if (m == i)
                continue;
              long double t = 1;
              for (int l = 0; l < n; ++l)
                if (l != i && l != m)
                  t *= (long double)x[q] - nodes[l];
              d += t;
            }
          V[q * n + i] = (double)(v / denom);
          D[q * n + i] = (double)(d / denom);
        }
  }

  // representative (smallest row-major index) of entry (i, j) of an n x n matrix that is
  // symmetric and invariant under (i, j) -> (n-1-i, n-1-j)
  constexpr int
  symmetric_entry(const int i, const int j, const int n)
  {
    const int a = i * n + j, b = j * n + i, c = (n - 1 - i) * n + (n - 1 - j), d = (n - 1 - j) * n + (n - 1 - i);
    const int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
  }

  // Eigenvectors of a pencil that is invariant under the reflection i -> m-1-i are even or odd
  // vectors.  If that is the case for Q[i*m + a] (m odd), reorder the eigenpairs so that the
  // (m+1)/2 even vectors come first, make the symmetry hold bitwise, and return true: kernels may
  // then apply Q in even-odd form.  The order of the eigenpairs has no meaning for the inverse.
  inline bool
  sort_eigenvectors_by_parity(const int m, std::vector<double> &Q, std::vector<double> &lam)
  {
    if (m % 2 == 0)
      return false;
    std::vector<int> even, odd;
    for (int a = 0; a < m; ++a)
      {
        double de = 0., dod = 0., nrm = 0.;
        for (int i = 0; i < m; ++i)
          {
            de  = std::max(de, std::fabs(Q[i * m + a] - Q[(m - 1 - i) * m + a]));
            dod = std::max(dod, std::fabs(Q[i * m + a] + Q[(m - 1 - i) * m + a]));
            nrm = std::max(nrm, std::fabs(Q[i * m + a]));
          }
        if (nrm == 0.)
          return false;
        if (de <= 1e-12 * nrm)
          even.push_back(a);
        else if (dod <= 1e-12 * nrm)
          odd.push_back(a);
        else
          return false;
      }
    if ((int)even.size() != (m + 1) / 2)
      return false;
    std::vector<int> order(even);
    order.insert(order.end(), odd.begin(), odd.end());
    const std::vector<double> Q0(Q), l0(lam);
    for (int a = 0; a < m; ++a)
      {
        const int    src  = order[a];
        const double sign = a < (int)even.size() ? 1. : -1.;
        lam[a]            = l0[src];
        for (int i = 0; i < m / 2; ++i)
          {
            const double x         = 0.5 * (Q0[i * m + src] + sign * Q0[(m - 1 - i) * m + src]);
            Q[i * m + a]           = x;
            Q[(m - 1 - i) * m + a] = sign * x;
          }
        Q[(m / 2) * m + a] = sign > 0. ? Q0[(m / 2) * m + src] : 0.;
      }
    return true;
  }

  struct Basis1D
  {
    int                 p, n1;
    std::vector<double> nodes, xq, wq;
    std::vector<double> N;  // [q][i] shape values at quadrature points
    std::vector<double> D;  // [q][i] shape derivatives at quadrature points
    std::vector<double> Dc; // [q][r] collocation derivative: d/dx of the Lagrange
                            // basis ON the quadrature points, at those points
    std::vector<double> M_ref, K_ref; // [i][j]
    // simultaneous diagonalisation of the reference matrices (filled by Basis1D::diagonalise, which
    // needs generalized_eigen): K_ref Q = M_ref Q diag(lam), Q^T M_ref Q = I, eigenvectors sorted by
    // parity; F = M_ref Q (row i, column a), so that  M_ref = F F^T  and  K_ref = F diag(lam) F^T
    std::vector<double> F_ref, lam_ref;
    void diagonalise();

    explicit Basis1D(const int degree)
      : p(degree)
      , n1(degree + 1)
    {
      nodes = gauss_lobatto(n1);
      gauss_legendre(n1, xq, wq);
      lagrange_eval(nodes, xq, N, D);
      std::vector<double> tmp;
      lagrange_eval(xq, xq, tmp, Dc);
      M_ref.assign(n1 * n1, 0.);
      K_ref.assign(n1 * n1, 0.);
      for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n1; ++j)
          {
            long double m = 0, k = 0;
            for (int q = 0; q < n1; ++q)
              {
                m += (long double)N[q * n1 + i] * N[q * n1 + j] * wq[q];
                k += (long double)D[q * n1 + i] * D[q * n1 + j] * wq[q];
              }
            M_ref[i * n1 + j] = (double)m;
            K_ref[i * n1 + j] = (double)k;
          }
      // The matrices are symmetric and, the nodes being symmetric about 1/2, invariant under
      // (i, j) -> (p - i, p - j).  Make both hold bitwise, so that kernels may keep the 9 (p = 4)
      // distinct entries only.
      for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n1; ++j)
          {
            const int e = symmetric_entry(i, j, n1);
            M_ref[i * n1 + j] = M_ref[e];
            K_ref[i * n1 + j] = K_ref[e];
          }
    }
  };

  // Generalized symmetric eigenproblem K q = lambda M q with Q^T M Q = I for the
  // small (m <= 13) dense 1-D patch matrices.  Rows/columns whose mass diagonal
  // is zero are "constrained" (Dirichlet or non-existing DoFs): they get a unit
  // diagonal during the solve and their eigenvector rows are zeroed afterwards,
  // which is what deal.II's TensorProductMatrixSymmetricSum does, so the FDM
  // inverse ignores such inputs and returns 0 there.
  // Output: lambda[a] ascending, Q[i*m + a] = component i of eigenvector a.
  inline void
  generalized_eigen(const int            m,
                    std::vector<double>  M,
                    std::vector<double>  K,
                    std::vector<double> &lambda,
                    std::vector<double> &Q)
  {
    std::vector<char> con(m, 0);
    for (int i = 0; i < m; ++i)
      if (M[i * m + i] == 0.)
        {
          con[i]       = 1;
          M[i * m + i] = 1.;
          K[i * m + i] = 1.;
        }
    // Cholesky M = L L^T
    std::vector<long double> L(m * m, 0.L);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j <= i; ++j)
        {
          long double s = M[i * m + j];
          for (int k = 0; k < j; ++k)
            s -= L[i * m + k] * L[j * m + k];
          if (i == j)
            {
              if (s <= 0)
                throw std::runtime_error("FDM setup: 1-D mass matrix not positive definite");
              L[i * m + i] = std::sqrt(s);
            }
          else
            L[i * m + j] = s / L[j * m + j];
        }
    // Linv
    std::vector<long double> Li(m * m, 0.L);
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < m; ++i)
        {
          long double s = (i == c) ? 1.L : 0.L;
          for (int k = 0; k < i; ++k)
            s -= L[i * m + k] * Li[k * m + c];
          Li[i * m + c] = s / L[i * m + i];
        }
    // C = Linv K Linv^T
    std::vector<long double> T(m * m, 0.L), C(m * m, 0.L);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j)
        {
          long double s = 0;
          for (int k = 0; k < m; ++k)
            s += Li[i * m + k] * K[k * m + j];
          T[i * m + j] = s;
        }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j)
        {
          long double s = 0;
          for (int k = 0; k < m; ++k)
            s += T[i * m + k] * Li[j * m + k];
          C[i * m + j] = s;
        }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < i; ++j)
        C[i * m + j] = C[j * m + i] = 0.5L * (C[i * m + j] + C[j * m + i]);
    // cyclic Jacobi
    std::vector<long double> V(m * m, 0.L);
    for (int i = 0; i < m; ++i)
      V[i * m + i] = 1.L;
    for (int sweep = 0; sweep < 100; ++sweep)
      {
        long double off = 0, diag = 0;
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < m; ++j)
            (i == j ? diag : off) += C[i * m + j] * C[i * m + j];
        if (off <= 1e-40L * diag || off == 0)
          break;
        for (int pq = 0; pq < m; ++pq)
          for (int qq = pq + 1; qq < m; ++qq)
            {
              const long double apq = C[pq * m + qq];
              if (apq == 0)
                continue;
              const long double theta = (C[qq * m + qq] - C[pq * m + pq]) / (2 * apq);
              const long double t =
                (theta >= 0 ? 1.L : -1.L) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
              const long double c = 1 / std::sqrt(t * t + 1), s = t * c;
              for (int k = 0; k < m; ++k)
                {
                  const long double akp = C[k * m + pq], akq = C[k * m + qq];
                  C[k * m + pq] = c * akp - s * akq;
                  C[k * m + qq] = s * akp + c * akq;
                }
              for (int k = 0; k < m; ++k)
                {
                  const long double apk = C[pq * m + k], aqk = C[qq * m + k];
                  C[pq * m + k] = c * apk - s * aqk;
                  C[qq * m + k] = s * apk + c * aqk;
                }
              for (int k = 0; k < m; ++k)
                {
                  const long double vkp = V[k * m + pq], vkq = V[k * m + qq];
                  V[k * m + pq] = c * vkp - s * vkq;
                  V[k * m + qq] = s * vkp + c * vkq;
                }
            }
      }
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i)
      order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return C[a * m + a] < C[b * m + b]; });
    lambda.assign(m, 0.);
    Q.assign(m * m, 0.);
    for (int a = 0; a < m; ++a)
      {
        const int src = order[a];
        lambda[a]     = (double)C[src * m + src];
        // q = Linv^T v
        for (int i = 0; i < m; ++i)
          {
            long double s = 0;
            for (int k = 0; k < m; ++k)
              s += Li[k * m + i] * V[k * m + src];
            Q[i * m + a] = con[i] ? 0. : (double)s;
          }
      }
  }

  inline void
  Basis1D::diagonalise()
  {
    std::vector<double> lam, Q;
    generalized_eigen(n1, M_ref, K_ref, lam, Q);
    if (!sort_eigenvectors_by_parity(n1, Q, lam))
      throw std::runtime_error("reference eigenvectors are not even / odd vectors");
    F_ref.assign(n1 * n1, 0.);
    for (int i = 0; i < n1; ++i)
      for (int a = 0; a < n1; ++a)
        {
          long double sum = 0;
          for (int j = 0; j < n1; ++j)
            sum += (long double)M_ref[i * n1 + j] * Q[j * n1 + a];
          F_ref[i * n1 + a] = (double)sum;
        }
    // make the parity of the columns of F hold bitwise
    const int ne = (n1 + 1) / 2;
    for (int a = 0; a < n1; ++a)
      {
        const double sign = a < ne ? 1. : -1.;
        for (int i = 0; i < n1 / 2; ++i)
          {
            const double x           = 0.5 * (F_ref[i * n1 + a] + sign * F_ref[(n1 - 1 - i) * n1 + a]);
            F_ref[i * n1 + a]           = x;
            F_ref[(n1 - 1 - i) * n1 + a] = sign * x;
          }
        if (sign < 0.)
          F_ref[(n1 / 2) * n1 + a] = 0.;
      }
    lam_ref = lam;
  }
} // namespace asmamd
