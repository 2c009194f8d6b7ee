// CPU restatement (C++, threaded) of one Chebyshev + FDM additive-Schwarz smoother
// application for Cartesian meshes, degree 4, n_overlap = 1, "post" weights.
//
// *** TEST INFRASTRUCTURE ONLY *** -- used by tests/ (checked against the numpy oracle)
// and by the cpu_baseline leg of bench.py; the product path never links or loads it.
//
// It follows the reference's sweep structure (not its source):
//   * cells are processed in SIMD batches of 8, one cell per lane, like deal.II's
//     VectorizedArray<double> cell batches (include/operator.h:1335-1351);
//   * A: gather -> sum-factorised evaluate(gradients) (interpolation to the Gauss points,
//     collocation derivative) -> Cartesian J^-1 J^-T detJ w -> integrate -> scatter-add
//     (do_cell_integral_local_base, include/operator.h:866-875);
//   * P: gather -> 3 sweeps with Q^T, scaling by 1/(lx+ly+lz), 3 sweeps with Q, weights,
//     scatter-add (cell_operation_normal, include/matrix_free.h:1023-1062);
//   * Chebyshev updates as in PreconditionChebyshev (deal.II) for the first-kind polynomial;
//   * threads take whole BLOCKS of cells (8 x 4 x 4) of one colour at a time (2x2x2 parity of the blocks: blocks of
//     a colour share no DoF), the cells of a block one after the other in SIMD batches of 8 consecutive cells along x:
//     a block's DoFs (76 KB per vector) stay in the core's cache, a batch row is 33 consecutive DoFs.  This is the
//     shared-memory counterpart of the reference's one-MPI-rank-per-core runs with its cell batches
//     (round 3; rounds 1-2 coloured single cells with stride 2: every gather a strided one, 1.3e8 DoF/s on 256 threads).
// Tables (1-D basis, eigen-decompositions, weights) are computed by oracle/asm_oracle.py
// and passed in, so this file contains the sweeps only.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <vector>

namespace
{
  constexpr int P = 4, N1 = 5, NE = 125, VL = 8;

  struct Problem
  {
    int           nc[3], nu[3], periodic[3];
    double        h[3];
    const double *N, *Dc, *wq;             // [q][i], [q][r], [q]
    const int *   inst[3];                 // [nc[d]]
    const double *eig;                     // [n_inst][30]
    const double *wpost[3];                // [nc[d]][5]
    // deformed meshes (per-cell data, cells numbered x fastest); nullptr on Cartesian meshes
    const double *q2_coeff = nullptr;      // [cell][27][3] monomial coefficients of the tri-quadratic map, index 9k+3j+i
    const double *cell_eig = nullptr;      // [cell][3][30]: Q (row i, col a) then lambda, per direction
    const double *cell_wpost = nullptr;    // [cell][125] post weights of the cell's patch
    const double *xq = nullptr;            // [5] quadrature points
  };
  inline long
  cell_index(const Problem &p, const int cx, const int cy, const int cz)
  {
    return cx + (long)p.nc[0] * (cy + (long)p.nc[1] * cz);
  }

  inline long
  dof(const Problem &p, int I, int J, int K)
  {
    if (p.periodic[0] && I == p.nu[0])
      I = 0;
    if (p.periodic[1] && J == p.nu[1])
      J = 0;
    if (p.periodic[2] && K == p.nu[2])
      K = 0;
    return I + (long)p.nu[0] * (J + (long)p.nu[1] * K);
  }
  inline bool
  constrained(const Problem &p, int d, int I)
  {
    return !p.periodic[d] && (I == 0 || I == p.nc[d] * P);
  }

  // out[a][.] = sum_i M[a*5+i] in[i][.] along direction d of a [5][5][5][VL] tile
  template <bool TRANSPOSE>
  inline void
  sweep(const double *M, const int d, double (*t)[VL])
  {
    const int stride = d == 0 ? 1 : (d == 1 ? N1 : N1 * N1);
    for (int o2 = 0; o2 < N1; ++o2)
      for (int o1 = 0; o1 < N1; ++o1)
        {
          const int base = d == 0 ? (o2 * N1 + o1) * N1 : (d == 1 ? o2 * N1 * N1 + o1 : o2 * N1 + o1);
          double    in[N1][VL], out[N1][VL];
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              in[i][v] = t[base + i * stride][v];
          for (int a = 0; a < N1; ++a)
            {
              for (int v = 0; v < VL; ++v)
                out[a][v] = 0.;
              for (int i = 0; i < N1; ++i)
                {
                  const double m = TRANSPOSE ? M[i * N1 + a] : M[a * N1 + i];
                  for (int v = 0; v < VL; ++v)
                    out[a][v] += m * in[i][v];
                }
            }
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              t[base + i * stride][v] = out[i][v];
        }
  }

  // same with a different matrix per lane: M[v] points to a 5x5 matrix Q[i][a]
  template <bool FWD>
  inline void
  sweep_q(const double *const *Q, const int d, double (*t)[VL])
  {
    const int stride = d == 0 ? 1 : (d == 1 ? N1 : N1 * N1);
    for (int o2 = 0; o2 < N1; ++o2)
      for (int o1 = 0; o1 < N1; ++o1)
        {
          const int base = d == 0 ? (o2 * N1 + o1) * N1 : (d == 1 ? o2 * N1 * N1 + o1 : o2 * N1 + o1);
          double    in[N1][VL], out[N1][VL];
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              in[i][v] = t[base + i * stride][v];
          for (int a = 0; a < N1; ++a)
            {
              for (int v = 0; v < VL; ++v)
                out[a][v] = 0.;
              for (int i = 0; i < N1; ++i)
                for (int v = 0; v < VL; ++v)
                  out[a][v] += (FWD ? Q[v][i * N1 + a] : Q[v][a * N1 + i]) * in[i][v];
            }
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              t[base + i * stride][v] = out[i][v];
        }
  }

  // same with the lanes' matrices interleaved: Qb[i * 5 + a][v]
  template <bool FWD>
  inline void
  sweep_qb(const double (*Qb)[VL], const int d, double (*t)[VL])
  {
    const int stride = d == 0 ? 1 : (d == 1 ? N1 : N1 * N1);
    for (int o2 = 0; o2 < N1; ++o2)
      for (int o1 = 0; o1 < N1; ++o1)
        {
          const int base = d == 0 ? (o2 * N1 + o1) * N1 : (d == 1 ? o2 * N1 * N1 + o1 : o2 * N1 + o1);
          double    in[N1][VL], out[N1][VL];
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              in[i][v] = t[base + i * stride][v];
          for (int a = 0; a < N1; ++a)
            {
              for (int v = 0; v < VL; ++v)
                out[a][v] = 0.;
              for (int i = 0; i < N1; ++i)
                for (int v = 0; v < VL; ++v)
                  out[a][v] += (FWD ? Qb[i * N1 + a][v] : Qb[a * N1 + i][v]) * in[i][v];
            }
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              t[base + i * stride][v] = out[i][v];
        }
  }

  struct Batch
  {
    int  n;            // active lanes
    int  c[VL][3];     // cell coordinates
    bool x_run = false; // the lanes are consecutive cells along x
  };

  // lanes are consecutive cells along x (same y, z) and the row does not wrap: 4 n + 1 consecutive DoFs per (j, k)
  inline bool
  contiguous_batch(const Problem &p, const Batch &bt)
  {
    return bt.x_run && (!p.periodic[0] || bt.c[0][0] + bt.n < p.nc[0]);
  }

  template <bool MASK>
  inline void
  gather(const Problem &p, const Batch &bt, const double *src, double (*t)[VL])
  {
    if (contiguous_batch(p, bt))
      {
        const int I0 = P * bt.c[0][0], nrow = P * bt.n + 1;
        for (int k = 0; k < N1; ++k)
          for (int j = 0; j < N1; ++j)
            {
              const int  J = P * bt.c[0][1] + j, K = P * bt.c[0][2] + k;
              const bool row_con = MASK && (constrained(p, 1, J) || constrained(p, 2, K));
              double     row[P * VL + 1];
              const double *s = src + dof(p, I0, J, K);
              for (int i = 0; i < nrow; ++i)
                row[i] = row_con ? 0. : s[i];
              for (int i = nrow; i < P * VL + 1; ++i)
                row[i] = 0.;
              if (MASK && constrained(p, 0, I0))
                row[0] = 0.;
              if (MASK && constrained(p, 0, I0 + nrow - 1))
                row[nrow - 1] = 0.;
              for (int i = 0; i < N1; ++i)
                for (int v = 0; v < VL; ++v)
                  t[(k * N1 + j) * N1 + i][v] = v < bt.n ? row[P * v + i] : 0.;
            }
        return;
      }
    for (int k = 0; k < N1; ++k)
      for (int j = 0; j < N1; ++j)
        for (int i = 0; i < N1; ++i)
          for (int v = 0; v < VL; ++v)
            {
              double val = 0.;
              if (v < bt.n)
                {
                  const int I = P * bt.c[v][0] + i, J = P * bt.c[v][1] + j, K = P * bt.c[v][2] + k;
                  if (!(MASK && (constrained(p, 0, I) || constrained(p, 1, J) || constrained(p, 2, K))))
                    val = src[dof(p, I, J, K)];
                }
              t[(k * N1 + j) * N1 + i][v] = val;
            }
  }

  inline void
  scatter_add(const Problem &p, const Batch &bt, const double (*t)[VL], double *dst, const double sign)
  {
    if (contiguous_batch(p, bt))
      {
        const int I0 = P * bt.c[0][0], nrow = P * bt.n + 1;
        for (int k = 0; k < N1; ++k)
          for (int j = 0; j < N1; ++j)
            {
              const int J = P * bt.c[0][1] + j, K = P * bt.c[0][2] + k;
              if (constrained(p, 1, J) || constrained(p, 2, K))
                continue;
              // the lanes' rows overlap in their end points: summed first, one update per DoF
              double row[P * VL + 1];
              for (int i = 0; i < P * VL + 1; ++i)
                row[i] = 0.;
              for (int v = 0; v < bt.n; ++v)
                for (int i = 0; i < N1; ++i)
                  row[P * v + i] += t[(k * N1 + j) * N1 + i][v];
              double *d = dst + dof(p, I0, J, K);
              for (int i = (constrained(p, 0, I0) ? 1 : 0); i < nrow - (constrained(p, 0, I0 + nrow - 1) ? 1 : 0); ++i)
                d[i] += sign * row[i];
            }
        return;
      }
    for (int k = 0; k < N1; ++k)
      for (int j = 0; j < N1; ++j)
        for (int i = 0; i < N1; ++i)
          for (int v = 0; v < bt.n; ++v)
            {
              const int I = P * bt.c[v][0] + i, J = P * bt.c[v][1] + j, K = P * bt.c[v][2] + k;
              if (!(constrained(p, 0, I) || constrained(p, 1, J) || constrained(p, 2, K)))
                dst[dof(p, I, J, K)] += sign * t[(k * N1 + j) * N1 + i][v];
            }
  }

  // dst += sign * A_cell src   (sum factorisation, 12 sweeps)
  void
  laplace_batch(const Problem &p, const Batch &bt, const double *src, double *dst, const double sign)
  {
    alignas(64) double u[NE][VL], g[3][NE][VL];
    gather<true>(p, bt, src, u);
    for (int d = 0; d < 3; ++d)
      sweep<false>(p.N, d, u); // values at the quadrature points
    const double det = p.h[0] * p.h[1] * p.h[2];
    for (int d = 0; d < 3; ++d)
      {
        std::memcpy(g[d], u, sizeof(u));
        sweep<false>(p.Dc, d, g[d]); // reference gradient component d
        for (int k = 0; k < N1; ++k)
          for (int j = 0; j < N1; ++j)
            for (int i = 0; i < N1; ++i)
              {
                const double f = det * p.wq[i] * p.wq[j] * p.wq[k] / (p.h[d] * p.h[d]);
                for (int v = 0; v < VL; ++v)
                  g[d][(k * N1 + j) * N1 + i][v] *= f;
              }
        sweep<true>(p.Dc, d, g[d]);
      }
    for (int e = 0; e < NE; ++e)
      for (int v = 0; v < VL; ++v)
        u[e][v] = g[0][e][v] + g[1][e][v] + g[2][e][v];
    for (int d = 0; d < 3; ++d)
      sweep<true>(p.N, d, u);
    scatter_add(p, bt, u, dst, sign);
  }


  // dst += sign * A_cell src on deformed cells ("quadratic geometry", include/operator.h:1035-1159): the Jacobian of
  // the tri-quadratic map is evaluated from the 27 coefficient vectors per cell by collapsing z, then y, then x (the
  // reference's order, :1101-1128), inverted per quadrature point (do_invert, :891-914) and applied as
  // J^-1 J^-T det w to the reference gradient (:1132-1152)
  void
  laplace_batch_deformed(const Problem &p, const Batch &bt, const double *src, double *dst, const double sign)
  {
    alignas(64) double u[NE][VL], g[3][NE][VL];
    gather<true>(p, bt, src, u);
    for (int d = 0; d < 3; ++d)
      sweep<false>(p.N, d, u);
    for (int d = 0; d < 3; ++d)
      {
        std::memcpy(g[d], u, sizeof(u));
        sweep<false>(p.Dc, d, g[d]);
      }
    for (int v = 0; v < VL; ++v)
      {
        const int     vc = v < bt.n ? v : 0;
        const double *c  = p.q2_coeff + cell_index(p, bt.c[vc][0], bt.c[vc][1], bt.c[vc][2]) * 81;
        for (int qz = 0; qz < N1; ++qz)
          {
            const double z = p.xq[qz];
            double       a[9][3], az[9][3]; // value / z-derivative, index 3j+i
            for (int ji = 0; ji < 9; ++ji)
              for (int e = 0; e < 3; ++e)
                {
                  const double c0 = c[(0 * 9 + ji) * 3 + e], c1 = c[(1 * 9 + ji) * 3 + e], c2 = c[(2 * 9 + ji) * 3 + e];
                  a[ji][e]  = c0 + z * (c1 + z * c2);
                  az[ji][e] = c1 + 2. * z * c2;
                }
            for (int qy = 0; qy < N1; ++qy)
              {
                const double y = p.xq[qy];
                double       b[3][3], by[3][3], bz[3][3]; // index i
                for (int i = 0; i < 3; ++i)
                  for (int e = 0; e < 3; ++e)
                    {
                      b[i][e]  = a[0 * 3 + i][e] + y * (a[1 * 3 + i][e] + y * a[2 * 3 + i][e]);
                      by[i][e] = a[1 * 3 + i][e] + 2. * y * a[2 * 3 + i][e];
                      bz[i][e] = az[0 * 3 + i][e] + y * (az[1 * 3 + i][e] + y * az[2 * 3 + i][e]);
                    }
                for (int qx = 0; qx < N1; ++qx)
                  {
                    const double x = p.xq[qx];
                    double       J[3][3];
                    for (int e = 0; e < 3; ++e)
                      {
                        J[0][e] = b[1][e] + 2. * x * b[2][e];
                        J[1][e] = by[0][e] + x * (by[1][e] + x * by[2][e]);
                        J[2][e] = bz[0][e] + x * (bz[1][e] + x * bz[2][e]);
                      }
                    // inverse (adjugate / determinant)
                    const double t00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], t10 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                                 t20 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
                    const double det = J[0][0] * t00 + J[0][1] * t10 + J[0][2] * t20, inv = 1. / det;
                    double       G[3][3];
                    G[0][0] = inv * t00;
                    G[0][1] = inv * (J[0][2] * J[2][1] - J[0][1] * J[2][2]);
                    G[0][2] = inv * (J[0][1] * J[1][2] - J[0][2] * J[1][1]);
                    G[1][0] = inv * t10;
                    G[1][1] = inv * (J[0][0] * J[2][2] - J[0][2] * J[2][0]);
                    G[1][2] = inv * (J[0][2] * J[1][0] - J[0][0] * J[1][2]);
                    G[2][0] = inv * t20;
                    G[2][1] = inv * (J[0][1] * J[2][0] - J[0][0] * J[2][1]);
                    G[2][2] = inv * (J[0][0] * J[1][1] - J[0][1] * J[1][0]);
                    const int    q  = (qz * N1 + qy) * N1 + qx;
                    const double dw = det * p.wq[qx] * p.wq[qy] * p.wq[qz];
                    const double gx = g[0][q][v], gy = g[1][q][v], gz = g[2][q][v];
                    const double t0 = (G[0][0] * gx + G[0][1] * gy + G[0][2] * gz) * dw, t1 = (G[1][0] * gx + G[1][1] * gy + G[1][2] * gz) * dw,
                                 t2 = (G[2][0] * gx + G[2][1] * gy + G[2][2] * gz) * dw;
                    g[0][q][v] = G[0][0] * t0 + G[1][0] * t1 + G[2][0] * t2;
                    g[1][q][v] = G[0][1] * t0 + G[1][1] * t1 + G[2][1] * t2;
                    g[2][q][v] = G[0][2] * t0 + G[1][2] * t1 + G[2][2] * t2;
                  }
              }
          }
      }
    for (int d = 0; d < 3; ++d)
      sweep<true>(p.Dc, d, g[d]);
    for (int e = 0; e < NE; ++e)
      for (int v = 0; v < VL; ++v)
        u[e][v] = g[0][e][v] + g[1][e][v] + g[2][e][v];
    for (int d = 0; d < 3; ++d)
      sweep<true>(p.N, d, u);
    scatter_add(p, bt, u, dst, sign);
  }

  // dst += P_cell^-1 src
  void
  fdm_batch(const Problem &p, const Batch &bt, const double *src, double *dst)
  {
    alignas(64) double t[NE][VL];
    gather<false>(p, bt, src, t);
    const double *Q[3][VL], *lam[3][VL], *w[3][VL];
    for (int d = 0; d < 3; ++d)
      for (int v = 0; v < VL; ++v)
        {
          const int vc = v < bt.n ? v : 0, c = bt.c[vc][d];
          if (p.cell_eig)
            Q[d][v] = p.cell_eig + (cell_index(p, bt.c[vc][0], bt.c[vc][1], bt.c[vc][2]) * 3 + d) * 30;
          else
            Q[d][v] = p.eig + (long)p.inst[d][c] * 30;
          lam[d][v] = Q[d][v] + 25;
          w[d][v]   = p.cell_eig ? nullptr : p.wpost[d] + (long)c * N1;
        }
    // the lanes' matrices side by side ([entry][lane]), as TensorProductMatrixSymmetricSum keeps them for a cell batch:
    // the sweeps then read one SIMD vector per entry
    alignas(64) double Qb[3][25][VL];
    for (int d = 0; d < 3; ++d)
      for (int e = 0; e < 25; ++e)
        for (int v = 0; v < VL; ++v)
          Qb[d][e][v] = Q[d][v][e];
    for (int d = 0; d < 3; ++d)
      sweep_qb<true>(Qb[d], d, t);
    for (int k = 0; k < N1; ++k)
      for (int j = 0; j < N1; ++j)
        for (int i = 0; i < N1; ++i)
          for (int v = 0; v < VL; ++v)
            t[(k * N1 + j) * N1 + i][v] /= lam[0][v][i] + (lam[1][v][j] + lam[2][v][k]);
    for (int d = 0; d < 3; ++d)
      sweep_qb<false>(Qb[d], d, t);
    if (p.cell_eig)
      {
        for (int v = 0; v < VL; ++v)
          {
            const int     vc = v < bt.n ? v : 0;
            const double *wc = p.cell_wpost + cell_index(p, bt.c[vc][0], bt.c[vc][1], bt.c[vc][2]) * NE;
            for (int e = 0; e < NE; ++e)
              t[e][v] *= wc[e];
          }
      }
    else
      for (int k = 0; k < N1; ++k)
        for (int j = 0; j < N1; ++j)
          for (int i = 0; i < N1; ++i)
            for (int v = 0; v < VL; ++v)
              t[(k * N1 + j) * N1 + i][v] *= w[0][v][i] * (w[1][v][j] * w[2][v][k]);
    scatter_add(p, bt, t, dst, 1.0);
  }

  // Persistent worker threads (the shared-memory stand-in for the reference's one MPI rank per core):
  // run(f) executes f(thread) on every thread and returns when all are done.  Thread t always works
  // on the t-th slice of the cell layers AND of the vector index range, so that the pages a thread
  // first touches (vectors are allocated uninitialised and zeroed in parallel) are the pages it
  // mostly works on afterwards (NUMA first-touch placement).
  class Pool
  {
  public:
    explicit Pool(const int n)
      : n_(n)
    {
      // ASM_ORACLE_CPUS = "c0,c1,...": thread t runs on CPU c[t mod size] (bench.py: one hardware thread per physical core,
      // spread over the host's core complexes, when the process may use fewer CPUs than it sees); the caller's thread gets its
      // own affinity back when the pool goes
      if (const char *e = std::getenv("ASM_ORACLE_CPUS"))
        for (const char *q = e; *q;)
          {
            char *     end = nullptr;
            const long c   = std::strtol(q, &end, 10);
            if (end == q)
              break;
            cpus_.push_back((int)c);
            q = *end ? end + 1 : end;
          }
      if (!cpus_.empty())
        {
          have_old_ = pthread_getaffinity_np(pthread_self(), sizeof(old_), &old_) == 0;
          pin(0);
        }
      for (int t = 1; t < n_; ++t)
        workers_.emplace_back([this, t]() {
          pin(t);
          loop(t);
        });
    }
    ~Pool()
    {
      if (!cpus_.empty() && have_old_)
        (void)pthread_setaffinity_np(pthread_self(), sizeof(old_), &old_);
      {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
        ++epoch_;
      }
      cv_.notify_all();
      for (auto &w : workers_)
        w.join();
    }
    int
    size() const
    {
      return n_;
    }
    void
    run(const std::function<void(int)> &f)
    {
      {
        std::lock_guard<std::mutex> lk(m_);
        job_     = &f;
        pending_ = n_ - 1;
        ++epoch_;
      }
      cv_.notify_all();
      f(0);
      std::unique_lock<std::mutex> lk(m_);
      done_.wait(lk, [this]() { return pending_ == 0; });
    }

  private:
    void
    pin(const int t) const
    {
      if (cpus_.empty())
        return;
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(cpus_[t % cpus_.size()], &set);
      (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    void
    loop(const int t)
    {
      long seen = 0;
      for (;;)
        {
          const std::function<void(int)> *job;
          {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&]() { return epoch_ != seen; });
            seen = epoch_;
            if (stop_)
              return;
            job = job_;
          }
          (*job)(t);
          {
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0)
              done_.notify_one();
          }
        }
    }
    int                             n_;
    std::vector<int>                cpus_;
    cpu_set_t                       old_;
    bool                            have_old_ = false;
    std::vector<std::thread>        workers_;
    std::mutex                      m_;
    std::condition_variable         cv_, done_;
    const std::function<void(int)> *job_ = nullptr;
    long                            epoch_ = 0;
    int                             pending_ = 0;
    bool                            stop_ = false;
  };

  // Blocks of BS[0] x BS[1] x BS[2] cells; colour = 2x2x2 parity of the block coordinates: no two blocks of a colour
  // share a DoF (a periodic direction needs an even number of blocks: its block size is reduced until it has).  The cells
  // of a block in SIMD batches of up to 8 consecutive cells along x, layer by layer; blocks ordered by z, so that thread
  // t's slice of the blocks is the t-th slab of the mesh (and of the vectors it first touched).
  struct ColourBlocks
  {
    std::vector<Batch>  batches;
    std::vector<size_t> block_begin; // batches [block_begin[b], block_begin[b + 1]) = block b
  };
  ColourBlocks
  colour_blocks(const Problem &p, const int colour)
  {
    int bs[3] = {VL, 4, 4}, nb[3];
    for (int d = 0; d < 3; ++d)
      {
        while (true)
          {
            nb[d] = (p.nc[d] + bs[d] - 1) / bs[d];
            if (!p.periodic[d] || nb[d] % 2 == 0 || bs[d] == 1)
              break;
            --bs[d];
          }
      }
    ColourBlocks cb;
    for (int bz = (colour >> 2) & 1; bz < nb[2]; bz += 2)
      for (int by = (colour >> 1) & 1; by < nb[1]; by += 2)
        for (int bx = colour & 1; bx < nb[0]; bx += 2)
          {
            cb.block_begin.push_back(cb.batches.size());
            for (int cz = bz * bs[2]; cz < std::min((bz + 1) * bs[2], p.nc[2]); ++cz)
              for (int cy = by * bs[1]; cy < std::min((by + 1) * bs[1], p.nc[1]); ++cy)
                {
                  Batch bt;
                  bt.n     = 0;
                  bt.x_run = true;
                  for (int cx = bx * bs[0]; cx < std::min((bx + 1) * bs[0], p.nc[0]); ++cx)
                    {
                      bt.c[bt.n][0] = cx;
                      bt.c[bt.n][1] = cy;
                      bt.c[bt.n][2] = cz;
                      if (++bt.n == VL)
                        {
                          cb.batches.push_back(bt);
                          bt.n = 0;
                        }
                    }
                  if (bt.n)
                    cb.batches.push_back(bt);
                }
          }
    cb.block_begin.push_back(cb.batches.size());
    return cb;
  }

  // run f(batch) over all cells, colour by colour, batches split over the pool's threads
  template <typename F>
  void
  cell_loop(Pool &pool, const ColourBlocks (&colours)[8], F &&f)
  {
    const int nt = pool.size();
    for (int colour = 0; colour < 8; ++colour)
      {
        const ColourBlocks &cb = colours[colour];
        const size_t        nb = cb.block_begin.size() - 1; // whole blocks per thread
        if (nb == 0)
          continue;
        pool.run([&](const int th) {
          for (size_t i = cb.block_begin[nb * th / nt]; i < cb.block_begin[nb * (th + 1) / nt]; ++i)
            f(cb.batches[i]);
        });
      }
  }

  // f(i0, i1) on the thread's slice of [0, n)
  template <typename F>
  void
  vector_loop(Pool &pool, const long n, F &&f)
  {
    const int nt = pool.size();
    pool.run([&](const int th) { f(n * th / nt, n * (th + 1) / nt); });
  }
} // namespace

namespace
{
  int
  cheb_step(const Problem &p, const int degree, const double theta, const double delta, double *x, const double *b, const int n_threads,
            const int reps)
  {
  const long n = (long)p.nu[0] * p.nu[1] * p.nu[2];
  Pool               pool(n_threads < 1 ? 1 : n_threads);
  ColourBlocks colours[8];
  for (int c = 0; c < 8; ++c)
    colours[c] = colour_blocks(p, c);
  // work vectors: uninitialised allocations, first touched by the thread that owns the slice
  const auto alloc = [&](const long count) { return static_cast<double *>(std::malloc(sizeof(double) * count)); };
  double *   xw = alloc(n), *bw = alloc(n), *r = alloc(n), *t = alloc(n), *xo = alloc(n);
  char *     con = static_cast<char *>(std::malloc(n));
  if (!xw || !bw || !r || !t || !xo || !con)
    return 2;
  const long nxy = (long)p.nu[0] * p.nu[1];
  vector_loop(pool, n, [&](const long i0, const long i1) {
    for (long i = i0; i < i1; ++i)
      {
        const int K = (int)(i / nxy), J = (int)((i % nxy) / p.nu[0]), I = (int)(i % p.nu[0]);
        con[i]      = constrained(p, 0, I) || constrained(p, 1, J) || constrained(p, 2, K);
        xw[i]       = x[i];
        bw[i]       = b[i];
        r[i] = t[i] = xo[i] = 0.;
      }
  });
  const auto residual = [&](const double *xx) {
    // r = b - A x, identity rows on constrained DoFs (matrix_free_internal.h:247-255)
    vector_loop(pool, n, [&](const long i0, const long i1) {
      for (long i = i0; i < i1; ++i)
        r[i] = con[i] ? bw[i] - xx[i] : bw[i];
    });
    if (p.q2_coeff)
      cell_loop(pool, colours, [&](const Batch &bt) { laplace_batch_deformed(p, bt, xx, r, -1.0); });
    else
      cell_loop(pool, colours, [&](const Batch &bt) { laplace_batch(p, bt, xx, r, -1.0); });
  };
  const auto precondition = [&]() {
    vector_loop(pool, n, [&](const long i0, const long i1) {
      for (long i = i0; i < i1; ++i)
        t[i] = con[i] ? r[i] : 0.;
    });
    cell_loop(pool, colours, [&](const Batch &bt) { fdm_batch(p, bt, r, t); });
  };
  for (int rep = 0; rep < reps; ++rep)
    {
      residual(xw);
      precondition();
      vector_loop(pool, n, [&](const long i0, const long i1) {
        for (long i = i0; i < i1; ++i)
          {
            xo[i] = xw[i];
            xw[i] += t[i] / theta;
          }
      });
      double       rhok  = delta / theta;
      const double sigma = theta / delta;
      for (int k = 1; k < degree; ++k)
        {
          const double rhokp = 1. / (2. * sigma - rhok);
          const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
          rhok = rhokp;
          residual(xw);
          precondition();
          vector_loop(pool, n, [&](const long i0, const long i1) {
            for (long i = i0; i < i1; ++i)
              {
                const double xn = xw[i] + f1 * (xw[i] - xo[i]) + f2 * t[i];
                xo[i]           = xw[i];
                xw[i]           = xn;
              }
          });
        }
    }
  vector_loop(pool, n, [&](const long i0, const long i1) {
    for (long i = i0; i < i1; ++i)
      x[i] = xw[i];
  });
  std::free(xw);
  std::free(bw);
  std::free(r);
  std::free(t);
  std::free(xo);
  std::free(con);
  return 0;
  }

  int
  fill_problem(Problem &p, const int *nc, const int *periodic, const double *h)
  {
    for (int d = 0; d < 3; ++d)
      {
        p.nc[d]       = nc[d];
        p.periodic[d] = periodic[d];
        p.nu[d]       = nc[d] * P + (periodic[d] ? 0 : 1);
        p.h[d]        = h[d];
        if (periodic[d] && (nc[d] % 2))
          return 1; // the 2x2x2 colouring of the blocks needs an even number of cells in periodic directions
      }
    return 0;
  }
} // namespace

extern "C" {

// x <- Chebyshev step (degree k, first kind) with the FDM preconditioner, `reps` times; Cartesian mesh
int
oracle_cpu_cheb_step(const int *nc, const int *periodic, const double *h, const double *N, const double *Dc, const double *wq,
                     const int *inst_x, const int *inst_y, const int *inst_z, const double *eig, const double *wpost_x,
                     const double *wpost_y, const double *wpost_z, const int degree, const double theta, const double delta,
                     double *x, const double *b, const int n_threads, const int reps)
{
  Problem p;
  if (fill_problem(p, nc, periodic, h))
    return 1;
  p.N        = N;
  p.Dc       = Dc;
  p.wq       = wq;
  p.inst[0]  = inst_x;
  p.inst[1]  = inst_y;
  p.inst[2]  = inst_z;
  p.eig      = eig;
  p.wpost[0] = wpost_x;
  p.wpost[1] = wpost_y;
  p.wpost[2] = wpost_z;
  return cheb_step(p, degree, theta, delta, x, b, n_threads, reps);
}

// the same on a deformed (tri-quadratic) Dirichlet mesh with per-cell eigen-decompositions and patch weights
int
oracle_cpu_cheb_step_deformed(const int *nc, const double *N, const double *Dc, const double *wq, const double *xq,
                              const double *q2_coeff, const double *cell_eig, const double *cell_wpost, const int degree,
                              const double theta, const double delta, double *x, const double *b, const int n_threads, const int reps)
{
  Problem      p;
  const int    periodic[3] = {0, 0, 0};
  const double h[3]        = {1., 1., 1.};
  if (fill_problem(p, nc, periodic, h))
    return 1;
  p.N          = N;
  p.Dc         = Dc;
  p.wq         = wq;
  p.xq         = xq;
  p.inst[0] = p.inst[1] = p.inst[2] = nullptr;
  p.eig                             = nullptr;
  p.wpost[0] = p.wpost[1] = p.wpost[2] = nullptr;
  p.q2_coeff                           = q2_coeff;
  p.cell_eig                           = cell_eig;
  p.cell_wpost                         = cell_wpost;
  return cheb_step(p, degree, theta, delta, x, b, n_threads, reps);
}
}
