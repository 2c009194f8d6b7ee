// Low-order sweeps (degree 1 and 2: 2^3 and 3^3 tiles) with ONE THREAD PER CELL, the whole tile in registers.
// These are the sweeps of the lower multigrid levels (p-coarsening 4 -> 2 -> 1 on the fine mesh, then the h levels):
// the thread-per-line kernels of kernels_generic.hip spend a workgroup barrier and an LDS pass per sweep on tiles
// of 8 or 27 values and take about as long per cell as at degree 4.
//
//   fdm_cell_kernel<M>     patch solve of ASPoissonPreconditioner::vmult_internal (include/matrix_free.h:1023-1162)
//                          for m = 2, 3: gather, pre weights, 6 sweeps + scaling, post weights; result to
//                          tmp[cell][m^3] (summed per DoF by fdm_gather_kernel) or added to dst with atomics
//   laplace_cell_kernel<P> LaplaceOperatorMatrixFree::do_cell_integral_global (include/operator.h:1335-1351) on
//                          Cartesian cells: for a box cell the quadrature form (exact Gauss(p+1), :866-875) equals
//                          hy hz/hx Mz My Kx + hx hz/hy Mz Ky Mx + hx hy/hz Kz My Mx with the 1-D reference matrices
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace asmamd
{
  namespace
  {
    constexpr int BLOCK = 256;

    // out[a] = sum_i Q[i*M + a] in[i] (FWD) or out[i] = sum_a Q[i*M + a] in[a], along direction D of u[(k*M + j)*M + i]
    template <int M, int D, bool FWD>
    __device__ __forceinline__ void
    sweep_cell(const double (&Q)[M * M], double (&u)[M * M * M])
    {
      constexpr int stride = D == 0 ? 1 : (D == 1 ? M : M * M);
#pragma unroll
      for (int o2 = 0; o2 < M; ++o2)
#pragma unroll
        for (int o1 = 0; o1 < M; ++o1)
          {
            const int base = D == 0 ? (o2 * M + o1) * M : (D == 1 ? o2 * M * M + o1 : o2 * M + o1);
            double    in[M], out[M];
#pragma unroll
            for (int i = 0; i < M; ++i)
              in[i] = u[base + i * stride];
#pragma unroll
            for (int a = 0; a < M; ++a)
              {
                double s = 0.;
#pragma unroll
                for (int i = 0; i < M; ++i)
                  s = fma(FWD ? Q[i * M + a] : Q[a * M + i], in[i], s);
                out[a] = s;
              }
#pragma unroll
            for (int i = 0; i < M; ++i)
              u[base + i * stride] = out[i];
          }
    }

    template <int M, bool STORE>
    __global__ void __launch_bounds__(BLOCK)
    fdm_cell_kernel(const DevGrid g, const DevFdm f, const double *__restrict__ src, double *__restrict__ dst)
    {
      constexpr int M2 = M * M, M3 = M * M * M, E = M2 + M;
      const long long ncells = (long long)g.nc[0] * g.nc[1] * g.nc[2];
      const long long cell   = xcd_block(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
      if (cell >= ncells)
        return;
      const int c3[3] = {(int)(cell % g.nc[0]), (int)((cell / g.nc[0]) % g.nc[1]), (int)(cell / ((long long)g.nc[0] * g.nc[1]))};
      const long long ce = (c3[0] + 1) + (g.nc[0] + 2) * ((c3[1] + 1) + (long long)(g.nc[1] + 2) * (c3[2] + 1));
      double    Q[3][M2], lam[3][M];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        {
          const int     id = f.tensor_inst ? f.inst[d][c3[d] + 1] : f.inst[d][ce];
          const double *e  = f.eig + (long long)id * E;
#pragma unroll
          for (int q = 0; q < M2; ++q)
            Q[d][q] = e[q];
#pragma unroll
          for (int q = 0; q < M; ++q)
            lam[d][q] = e[M2 + q];
        }
      const int o[3] = {f.origin_mul * c3[0] + f.origin_add, f.origin_mul * c3[1] + f.origin_add, f.origin_mul * c3[2] + f.origin_add};
      long long off[3][M];
      bool      ex[3][M];
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < M; ++i)
          {
            ex[d][i]  = g.exists(d, o[d] + i);
            off[d][i] = ex[d][i] ? g.store_index(d, o[d] + i) * g.stride[d] : 0;
          }
      double u[M3];
#pragma unroll
      for (int k = 0; k < M; ++k)
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
          for (int i = 0; i < M; ++i)
            {
              const double w = f.w_pre[0][(c3[0] + 1) * M + i] * f.w_pre[1][(c3[1] + 1) * M + j] * f.w_pre[2][(c3[2] + 1) * M + k];
              double       v = 0.;
              if (ex[0][i] && ex[1][j] && ex[2][k] && w != 0.)
                v = src[off[0][i] + off[1][j] + off[2][k]] * w;
              u[(k * M + j) * M + i] = v;
            }
      sweep_cell<M, 2, true>(Q[2], u);
      sweep_cell<M, 1, true>(Q[1], u);
      sweep_cell<M, 0, true>(Q[0], u);
#pragma unroll
      for (int c = 0; c < M; ++c)
#pragma unroll
        for (int b = 0; b < M; ++b)
#pragma unroll
          for (int a = 0; a < M; ++a)
            {
              // reciprocal with two Newton steps (full double precision)
              const double sm = lam[0][a] + (lam[1][b] + lam[2][c]);
              double       r  = __builtin_amdgcn_rcp(sm);
              r               = fma(fma(-sm, r, 1.0), r, r);
              r               = fma(fma(-sm, r, 1.0), r, r);
              u[(c * M + b) * M + a] *= r;
            }
      sweep_cell<M, 0, false>(Q[0], u);
      sweep_cell<M, 1, false>(Q[1], u);
      sweep_cell<M, 2, false>(Q[2], u);
#pragma unroll
      for (int k = 0; k < M; ++k)
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
          for (int i = 0; i < M; ++i)
            {
              const double w  = f.w_post[0][(c3[0] + 1) * M + i] * f.w_post[1][(c3[1] + 1) * M + j] * f.w_post[2][(c3[2] + 1) * M + k];
              const bool   ok = ex[0][i] && ex[1][j] && ex[2][k] && w != 0.;
              if (STORE)
                dst[((k * M + j) * M + i) * ncells + cell] = ok ? u[(k * M + j) * M + i] * w : 0.; // tmp[entry][cell]
              else if (ok)
                unsafeAtomicAdd(&dst[off[0][i] + off[1][j] + off[2][k]], u[(k * M + j) * M + i] * w);
            }
    }

    // y = A u along direction D of u[(k*N + j)*N + i] with a symmetric N x N matrix
    template <int N, int D>
    __device__ __forceinline__ void
    apply_sym(const double (&A)[9], const double (&u)[N * N * N], double (&y)[N * N * N])
    {
      constexpr int stride = D == 0 ? 1 : (D == 1 ? N : N * N);
#pragma unroll
      for (int o2 = 0; o2 < N; ++o2)
#pragma unroll
        for (int o1 = 0; o1 < N; ++o1)
          {
            const int base = D == 0 ? (o2 * N + o1) * N : (D == 1 ? o2 * N * N + o1 : o2 * N + o1);
#pragma unroll
            for (int a = 0; a < N; ++a)
              {
                double s = 0.;
#pragma unroll
                for (int i = 0; i < N; ++i)
                  s = fma(A[a * N + i], u[base + i * stride], s);
                y[base + a * stride] = s;
              }
          }
    }

    template <int P, bool DIAG>
    __global__ void __launch_bounds__(BLOCK)
    laplace_cell_kernel(const DevGrid g, const DevOp op, const double *__restrict__ src, double *__restrict__ dst, const double sign)
    {
      constexpr int N = P + 1, N3 = N * N * N;
      // One wave = a tile of 8 x 8 cells of one z layer (lane = 8 ly + lx): the columns / rows neighbouring cells share are
      // merged through lane shuffles before the scatter.  Lanes outside the mesh run along (they take part in the
      // shuffles) and add nothing.
      const int       lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
      const int       ntx = (g.nc[0] + 7) / 8, nty = (g.nc[1] + 7) / 8;
      const long long tile = xcd_block(blockIdx.x, gridDim.x) * (BLOCK / 64) + (threadIdx.x >> 6);
      int             c3[3] = {(int)(tile % ntx) * 8 + lx, (int)((tile / ntx) % nty) * 8 + ly, (int)(tile / ((long long)ntx * nty))};
      const bool      active = c3[0] < g.nc[0] && c3[1] < g.nc[1] && c3[2] < g.nc[2];
      if (!active)
        c3[0] = c3[1] = c3[2] = 0;
      long long off[3][N];
      bool      ok[3][N]; // the DoF exists and is not constrained
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < N; ++i)
          {
            const int I = P * c3[d] + i;
            ok[d][i]    = active && g.exists(d, I) && !g.dirichlet(d, I);
            off[d][i]   = ok[d][i] ? g.store_index(d, I) * g.stride[d] : 0;
          }
      const double hx = op.h[0][c3[0] + 1], hy = op.h[1][c3[1] + 1], hz = op.h[2][c3[2] + 1];
      const double ax = hy * hz / hx, ay = hx * hz / hy, az = hx * hy / hz;
      double       M[9], K[9];
#pragma unroll
      for (int q = 0; q < N * N; ++q)
        {
          M[q] = op.M_small[q];
          K[q] = op.K_small[q];
        }
      if (DIAG)
        {
#pragma unroll
          for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
              for (int i = 0; i < N; ++i)
                if (ok[0][i] && ok[1][j] && ok[2][k])
                  {
                    const double mi = M[i * N + i], mj = M[j * N + j], mk = M[k * N + k];
                    const double d  = ax * mk * mj * K[i * N + i] + ay * mk * K[j * N + j] * mi + az * K[k * N + k] * mj * mi;
                    unsafeAtomicAdd(&dst[off[0][i] + off[1][j] + off[2][k]], d);
                  }
          return;
        }
      double u[N3];
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int i = 0; i < N; ++i)
            u[(k * N + j) * N + i] = (ok[0][i] && ok[1][j] && ok[2][k]) ? src[off[0][i] + off[1][j] + off[2][k]] : 0.;
      // out = Mz (ax My Kx u + ay Ky Mx u) + az Kz My Mx u
      double xm[N3], xk[N3], t[N3], acc[N3];
      apply_sym<N, 0>(M, u, xm);
      apply_sym<N, 0>(K, u, xk);
      apply_sym<N, 1>(M, xk, acc); // My Kx u
      apply_sym<N, 1>(K, xm, t);   // Ky Mx u
#pragma unroll
      for (int e = 0; e < N3; ++e)
        acc[e] = ax * acc[e] + ay * t[e];
      apply_sym<N, 2>(M, acc, xk); // Mz (...)
      apply_sym<N, 1>(M, xm, t);   // My Mx u
      apply_sym<N, 2>(K, t, acc);  // Kz My Mx u
      // Merge inside the wave: the last column (row) of a cell and the first column (row) of its right (upper) neighbour
      // - the next lane (the lane 8 further) - are the same DoFs.  The left (lower) lane hands them over and skips those
      // atomics: 12 instead of 27 of them per cell at degree 2, and none of the back-to-back updates of one address by
      // neighbouring lanes that serialise in the L2 (measured on 80^3 cells at degree 2: the scatter took 127 of the
      // sweep's 160 us with one thread per cell in x order).
      const bool give_right = lx < 7 && c3[0] + 1 < g.nc[0] && active, take_left = lx > 0 && active;
      const bool give_up    = ly < 7 && c3[1] + 1 < g.nc[1] && active, take_down = ly > 0 && active;
      double     out[N3];
#pragma unroll
      for (int e = 0; e < N3; ++e)
        out[e] = active ? xk[e] + az * acc[e] : 0.;
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j)
          {
            const double left = __shfl_up(out[(k * N + j) * N + N - 1], 1, 64);
            if (take_left)
              out[(k * N + j) * N] += left;
          }
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i)
          {
            const double down = __shfl_up(out[(k * N + N - 1) * N + i], 8, 64);
            if (take_down)
              out[(k * N) * N + i] += down;
          }
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
          for (int i = 0; i < N; ++i)
            {
              const bool given = (i == N - 1 && give_right) || (j == N - 1 && give_up);
              if (!given && ok[0][i] && ok[1][j] && ok[2][k])
                unsafeAtomicAdd(&dst[off[0][i] + off[1][j] + off[2][k]], sign * out[(k * N + j) * N + i]);
            }
    }
  } // namespace

  bool
  fdm_cell(const DevGrid &g, const DevFdm &f, const double *src, double *dst, const bool store, hipStream_t st)
  {
    const long long ncells = (long long)g.nc[0] * g.nc[1] * g.nc[2];
    const unsigned  blocks = (unsigned)((ncells + BLOCK - 1) / BLOCK);
    if (f.m == 2)
      {
        if (store)
          hipLaunchKernelGGL((fdm_cell_kernel<2, true>), dim3(blocks), dim3(BLOCK), 0, st, g, f, src, dst);
        else
          hipLaunchKernelGGL((fdm_cell_kernel<2, false>), dim3(blocks), dim3(BLOCK), 0, st, g, f, src, dst);
        return true;
      }
    if (f.m == 3)
      {
        if (store)
          hipLaunchKernelGGL((fdm_cell_kernel<3, true>), dim3(blocks), dim3(BLOCK), 0, st, g, f, src, dst);
        else
          hipLaunchKernelGGL((fdm_cell_kernel<3, false>), dim3(blocks), dim3(BLOCK), 0, st, g, f, src, dst);
        return true;
      }
    return false;
  }

  bool
  laplace_cell(const DevGrid &g, const DevOp &op, const double *src, double *dst, const bool diag, hipStream_t st, const double sign)
  {
    if (!op.cartesian || g.p > 2)
      return false;
    const long long tiles  = (long long)((g.nc[0] + 7) / 8) * ((g.nc[1] + 7) / 8) * g.nc[2]; // 8 x 8 cells per wave
    const unsigned  blocks = (unsigned)((tiles + BLOCK / 64 - 1) / (BLOCK / 64));
    if (g.p == 1)
      {
        if (diag)
          hipLaunchKernelGGL((laplace_cell_kernel<1, true>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
        else
          hipLaunchKernelGGL((laplace_cell_kernel<1, false>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
      }
    else
      {
        if (diag)
          hipLaunchKernelGGL((laplace_cell_kernel<2, true>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
        else
          hipLaunchKernelGGL((laplace_cell_kernel<2, false>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
      }
    return true;
  }
} // namespace asmamd
