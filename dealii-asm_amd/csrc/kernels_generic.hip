// Generic (any degree / any patch box) HIP kernels for gfx950.
//
//   fdm_generic_kernel<M>      ASPoissonPreconditioner::vmult_internal cell operation
//                              (include/matrix_free.h:1023-1162): gather patch box ->
//                              pre weights -> 6 one-dimensional eigenvector sweeps +
//                              diagonal scaling (TensorProductMatrixSymmetricSum::
//                              apply_inverse, deal.II) -> post weights -> scatter-add
//   laplace_generic_kernel<P>  LaplaceOperatorMatrixFree::do_cell_integral_global
//                              (include/operator.h:1335-1351) with the Cartesian or the
//                              "quadratic geometry" Jacobians (:1035-1159)
//
// One thread owns one line of the patch tile; the tile lives in LDS between the
// sweeps; the line being transformed lives in registers.  Neighbouring patches
// overlap, so the scatter uses hardware fp64 atomics (global_atomic_add_f64).
// These kernels are the general path (every degree 1..7, n_overlap, vertex-star
// patches); the tiled kernels in kernels_tiled.hip are the fast path for the
// headline configuration.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace asmamd
{
  namespace
  {
    constexpr int BLOCK = 256;

    template <int M>
    struct FdmCfg
    {
      static constexpr int M2  = M * M;
      static constexpr int M3  = M * M * M;
      static constexpr int CPB = (BLOCK / M2) > 0 ? (BLOCK / M2) : 1; // cells per block
      static constexpr int E   = M2 + M;                              // eigen data per direction
    };

    // out[a] = sum_i Q[i][a] in[i]   (forward, Q^T)
    // a block-uniform eigenvector matrix in global memory: scalar loads, scalar operands of the FMAs
    // (no LDS traffic for the matrix entries)
    struct UniformMatrix
    {
      const __attribute__((address_space(4))) double *p;
      __device__ __forceinline__ explicit UniformMatrix(const double *q)
        : p((const __attribute__((address_space(4))) double *)q)
      {}
      __device__ __forceinline__ double operator[](const int e) const { return p[e]; }
    };
    // even-odd forms for odd M and a matrix whose first (M+1)/2 columns are even and the others odd
    // vectors (numerics.h: sort_eigenvectors_by_parity): about half the products
    template <int M, typename QT>
    __device__ __forceinline__ void
    sweep_fwd_eo(const QT &Q, double (&u)[M])
    {
      constexpr int H = M / 2, NE = H + 1;
      double        e[NE], o[H], t[M];
#pragma unroll
      for (int i = 0; i < H; ++i)
        {
          e[i] = u[i] + u[M - 1 - i];
          o[i] = u[i] - u[M - 1 - i];
        }
      e[H] = u[H];
#pragma unroll
      for (int a = 0; a < NE; ++a)
        {
          double s = Q[0 * M + a] * e[0];
#pragma unroll
          for (int i = 1; i < NE; ++i)
            s = fma(Q[i * M + a], e[i], s);
          t[a] = s;
        }
#pragma unroll
      for (int a = NE; a < M; ++a)
        {
          double s = Q[0 * M + a] * o[0];
#pragma unroll
          for (int i = 1; i < H; ++i)
            s = fma(Q[i * M + a], o[i], s);
          t[a] = s;
        }
#pragma unroll
      for (int a = 0; a < M; ++a)
        u[a] = t[a];
    }
    template <int M, typename QT>
    __device__ __forceinline__ void
    sweep_bwd_eo(const QT &Q, double (&u)[M])
    {
      constexpr int H = M / 2, NE = H + 1;
      double        E[NE], O[H];
#pragma unroll
      for (int i = 0; i < NE; ++i)
        {
          double s = Q[i * M + 0] * u[0];
#pragma unroll
          for (int a = 1; a < NE; ++a)
            s = fma(Q[i * M + a], u[a], s);
          E[i] = s;
        }
#pragma unroll
      for (int i = 0; i < H; ++i)
        {
          double s = Q[i * M + NE] * u[NE];
#pragma unroll
          for (int a = NE + 1; a < M; ++a)
            s = fma(Q[i * M + a], u[a], s);
          O[i] = s;
        }
#pragma unroll
      for (int i = 0; i < H; ++i)
        {
          u[i]         = E[i] + O[i];
          u[M - 1 - i] = E[i] - O[i];
        }
      u[H] = E[H];
    }

    template <int M, typename QT>
    __device__ __forceinline__ void
    sweep_fwd(const QT &Q, double (&u)[M])
    {
      double t[M];
#pragma unroll
      for (int a = 0; a < M; ++a)
        t[a] = 0.;
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int a = 0; a < M; ++a)
          t[a] = fma(Q[i * M + a], u[i], t[a]);
#pragma unroll
      for (int a = 0; a < M; ++a)
        u[a] = t[a];
    }

    // out[i] = sum_a Q[i][a] in[a]   (backward, Q)
    template <int M, typename QT>
    __device__ __forceinline__ void
    sweep_bwd(const QT &Q, double (&u)[M])
    {
      double t[M];
#pragma unroll
      for (int i = 0; i < M; ++i)
        {
          double s = 0.;
#pragma unroll
          for (int a = 0; a < M; ++a)
            s = fma(Q[i * M + a], u[a], s);
          t[i] = s;
        }
#pragma unroll
      for (int i = 0; i < M; ++i)
        u[i] = t[i];
    }

    template <int M, bool FWD>
    __device__ __forceinline__ void
    sweep_dir(const int mode, const double *__restrict__ Qlds, const UniformMatrix &U, double (&u)[M])
    {
      if constexpr ((M % 2) == 1)
        if (mode == 2)
          {
            if (FWD)
              sweep_fwd_eo<M>(U, u);
            else
              sweep_bwd_eo<M>(U, u);
            return;
          }
      if (mode >= 1)
        {
          if (FWD)
            sweep_fwd<M>(U, u);
          else
            sweep_bwd<M>(U, u);
        }
      else
        {
          if (FWD)
            sweep_fwd<M>(Qlds, u);
          else
            sweep_bwd<M>(Qlds, u);
        }
    }

    // STORE = false: contributions are added to dst with atomics (dst zeroed by the caller);
    // STORE = true:  the weighted patch result goes to tmp[cell][k][j][i] (every entry, zero where the
    //                entry does not contribute) and fdm_gather_kernel sums the patches per DoF
    template <int M, bool STORE>
    __global__ void __launch_bounds__(BLOCK)
    fdm_generic_kernel(const DevGrid g, const DevFdm f, const FdmFused fu, const double *__restrict__ src, double *__restrict__ dst)
    {
      using C = FdmCfg<M>;
      __shared__ double T[C::CPB][C::M3];
      __shared__ double Qs[C::CPB][3][C::E];

      const int  tid    = threadIdx.x;
      const int  lc     = tid / C::M2;
      const int  lt     = tid % C::M2;
      const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
      // M = 7 with a slice-per-lane box (kernels_slice7.hip): this kernel takes the other cells, from a compact list
      const bool listed = M == 7 && f.s7_on && (STORE || fu.direct == 2);
      const int  blk    = (int)xcd_block(blockIdx.x, gridDim.x);
      const int  slot   = blk * C::CPB + lc;
      const bool active = (lc < C::CPB) && (slot < (listed ? f.s7_n_others : ncells));
      const int  cell   = active ? (listed ? f.s7_others[slot] : slot) : 0;

      int c3[3] = {0, 0, 0}, ids[3] = {0, 0, 0};
      if (active)
        {
          c3[0] = cell % g.nc[0];
          c3[1] = (cell / g.nc[0]) % g.nc[1];
          c3[2] = cell / (g.nc[0] * g.nc[1]);
        }
      if (active)
        {
          // eigen data of the three directions -> LDS
          const int ce = (c3[0] + 1) + (g.nc[0] + 2) * ((c3[1] + 1) + (g.nc[1] + 2) * (c3[2] + 1));
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
              const int     id = f.tensor_inst ? f.inst[d][c3[d] + 1] : f.inst[d][ce];
              const double *e  = f.eig + (long long)id * C::E;
              ids[d]           = id;
              Qs[lc][d][lt]    = e[lt];
              if (lt < M)
                Qs[lc][d][C::M2 + lt] = e[C::M2 + lt];
            }
        }
      // Do all cells of the block share the instance of a direction (interior of a tensor-product mesh)?
      // Then its matrix is read with scalar loads instead of 49 LDS reads per line.
      int id0[3] = {0, 0, 0}, qmode[3] = {0, 0, 0};
      {
        const int cell0 = listed ? f.s7_others[blk * C::CPB] : blk * C::CPB; // first cell of the block: exists
        const int c0[3] = {cell0 % g.nc[0], (cell0 / g.nc[0]) % g.nc[1], cell0 / (g.nc[0] * g.nc[1])};
        const int ce0   = (c0[0] + 1) + (g.nc[0] + 2) * ((c0[1] + 1) + (g.nc[1] + 2) * (c0[2] + 1));
#pragma unroll
        for (int d = 0; d < 3; ++d)
          {
            id0[d] = __builtin_amdgcn_readfirstlane(f.tensor_inst ? f.inst[d][c0[d] + 1] : f.inst[d][ce0]);
            if (__syncthreads_and(!active || ids[d] == id0[d]))
              qmode[d] = (f.eig_sym != nullptr && f.eig_sym[id0[d]] != 0) ? 2 : 1;
          }
      }
      const UniformMatrix U0(f.eig + (long long)id0[0] * C::E), U1(f.eig + (long long)id0[1] * C::E), U2(f.eig + (long long)id0[2] * C::E);

      // thread (b, a) <-> lt = b*M + a
      const int a = lt % M, b = lt / M;
      const int o0 = f.origin_mul * c3[0] + f.origin_add, o1 = f.origin_mul * c3[1] + f.origin_add,
                o2 = f.origin_mul * c3[2] + f.origin_add;

      double u[M];
      // ---- gather: thread (j=b, i=a) reads the column along z --------------
      if (active)
        {
          const int    I = o0 + a, J = o1 + b;
          const bool   exy = g.exists(0, I) && g.exists(1, J);
          const double wxy = f.w_pre[0][(c3[0] + 1) * M + a] * f.w_pre[1][(c3[1] + 1) * M + b];
          const long long base = exy ? (g.store_index(0, I) * g.stride[0] + g.store_index(1, J) * g.stride[1]) : 0;
#pragma unroll
          for (int k = 0; k < M; ++k)
            {
              const int    K = o2 + k;
              const double w = wxy * f.w_pre[2][(c3[2] + 1) * M + k];
              double       v = 0.;
              if (exy && g.exists(2, K) && w != 0.)
                v = src[base + g.store_index(2, K) * g.stride[2]] * w;
              u[k] = v;
            }
        }
      __syncthreads(); // Qs visible
      if (active)
        {
          sweep_dir<M, true>(qmode[2], Qs[lc][2], U2, u); // z forward
#pragma unroll
          for (int k = 0; k < M; ++k)
            T[lc][k * C::M2 + b * M + a] = u[k];
        }
      __syncthreads();
      if (active)
        {
          // thread (k=b, i=a): column along y
#pragma unroll
          for (int j = 0; j < M; ++j)
            u[j] = T[lc][b * C::M2 + j * M + a];
          sweep_dir<M, true>(qmode[1], Qs[lc][1], U1, u);
#pragma unroll
          for (int j = 0; j < M; ++j)
            T[lc][b * C::M2 + j * M + a] = u[j];
        }
      __syncthreads();
      if (active)
        {
          // thread (k=b, j=a): row along x; forward, scale, backward
#pragma unroll
          for (int i = 0; i < M; ++i)
            u[i] = T[lc][b * C::M2 + a * M + i];
          sweep_dir<M, true>(qmode[0], Qs[lc][0], U0, u);
          const double lyz = Qs[lc][1][C::M2 + a] + Qs[lc][2][C::M2 + b];
#pragma unroll
          for (int i = 0; i < M; ++i)
            {
              // reciprocal with two Newton steps (full double precision, a fifth of an IEEE division)
              const double sm = Qs[lc][0][C::M2 + i] + lyz;
              double       r  = __builtin_amdgcn_rcp(sm);
              r               = fma(fma(-sm, r, 1.0), r, r);
              r               = fma(fma(-sm, r, 1.0), r, r);
              u[i] *= r;
            }
          sweep_dir<M, false>(qmode[0], Qs[lc][0], U0, u);
#pragma unroll
          for (int i = 0; i < M; ++i)
            T[lc][b * C::M2 + a * M + i] = u[i];
        }
      __syncthreads();
      if (active)
        {
          // thread (k=b, i=a): column along y, backward
#pragma unroll
          for (int j = 0; j < M; ++j)
            u[j] = T[lc][b * C::M2 + j * M + a];
          sweep_dir<M, false>(qmode[1], Qs[lc][1], U1, u);
#pragma unroll
          for (int j = 0; j < M; ++j)
            T[lc][b * C::M2 + j * M + a] = u[j];
        }
      __syncthreads();
      if (active)
        {
          // thread (j=b, i=a): column along z, backward, weights, scatter-add
#pragma unroll
          for (int k = 0; k < M; ++k)
            u[k] = T[lc][k * C::M2 + b * M + a];
          sweep_dir<M, false>(qmode[2], Qs[lc][2], U2, u);
          const int    I = o0 + a, J = o1 + b;
          const bool   exy = g.exists(0, I) && g.exists(1, J);
          const double wxy = f.w_post[0][(c3[0] + 1) * M + a] * f.w_post[1][(c3[1] + 1) * M + b];
          if (STORE)
            {
              double *t = dst + (long long)cell * C::M3 + b * M + a;
#pragma unroll
              for (int k = 0; k < M; ++k)
                {
                  const double w = wxy * f.w_post[2][(c3[2] + 1) * M + k];
                  t[k * C::M2]   = (exy && g.exists(2, o2 + k) && w != 0.) ? u[k] * w : 0.;
                }
            }
          else if (exy && wxy != 0.)
            {
              const long long base = g.store_index(0, I) * g.stride[0] + g.store_index(1, J) * g.stride[1];
#pragma unroll
              for (int k = 0; k < M; ++k)
                {
                  const int    K = o2 + k;
                  const double w = wxy * f.w_post[2][(c3[2] + 1) * M + k];
                  if (g.exists(2, K) && w != 0.)
                    {
                      const long long o = base + g.store_index(2, K) * g.stride[2];
                      if (!fu.direct)
                        unsafeAtomicAdd(&dst[o], u[k] * w);
                      else if (fu.direct == 2)
                        {
                          // beside fdm_slice7_acc: dst holds the other terms of the update, constrained DoFs their final value
                          if (fu.mode == PC_PLAIN)
                            unsafeAtomicAdd(&dst[o], u[k] * w);
                          else if (!(g.dirichlet(0, I) || g.dirichlet(1, J) || g.dirichlet(2, K)))
                            unsafeAtomicAdd(&dst[o], fu.f2 * (u[k] * w));
                        }
                      else if (!(g.dirichlet(0, I) || g.dirichlet(1, J) || g.dirichlet(2, K)))
                        {
                          // (constrained DoFs are left alone: vec_constrained_fused sets them from the unit row and must
                          //  still find x_old there when dst is x_old)
                          // the only patch that contributes to this DoF: store the updated value (FdmFused::direct)
                          double t = u[k] * w;
                          if (fu.mode == PC_FIRST)
                            t = fu.f2 * t;
                          else if (fu.mode == PC_STEP1)
                            t = fu.f1 * fu.x[o] + fu.f2 * t;
                          else if (fu.mode == PC_UPDATE)
                            {
                              const double xv = fu.x[o];
                              t               = xv + fu.f1 * (xv - fu.x_old[o]) + fu.f2 * t;
                            }
                          dst[o] = t;
                        }
                    }
                }
            }
        }
    }

    // Sum of the patch results per DoF (second half of the scatter-add of include/matrix_free.h:
    // 1023-1162 without atomics): one thread per stored DoF (ghost planes included: they carry this
    // rank's contributions to DoFs of its neighbours until compress(add)); along every direction the
    // DoF lies in the patches of at most ceil(m / p) cells.
    struct GatherDir
    {
      int n;
      int cell[4], idx[4];
    };
    __device__ __forceinline__ void
    gather_candidates(const DevGrid &g, const DevFdm &f, const int d, const int I, GatherDir &out)
    {
      const int np = g.nc[d] * g.p, m = f.m;
      out.n        = 0;
      for (int s = (g.wrap[d] ? -1 : 0); s <= (g.wrap[d] ? 1 : 0); ++s)
        {
          const int Is = I + s * np, t = Is - f.origin_add;
          // largest cell whose patch origin is not beyond Is, then downwards while the patch still covers Is
          int c = t >= 0 ? t / f.origin_mul : -((-t + f.origin_mul - 1) / f.origin_mul);
          for (; Is - (f.origin_mul * c + f.origin_add) < m; --c)
            if (c >= 0 && c < g.nc[d] && out.n < 4)
              {
                out.cell[out.n] = c;
                out.idx[out.n]  = Is - (f.origin_mul * c + f.origin_add);
                ++out.n;
              }
        }
    }

    // fu: the vector update PreconditionChebyshev hands over as post-operation (TiledMode PC_*), applied while the
    // sum is stored (one rank only: with several ranks compress(add) comes between the sum and the update)
    __global__ void __launch_bounds__(BLOCK)
    fdm_gather_kernel(const DevGrid g, const DevFdm f, const FdmFused fu, const double *__restrict__ src, const double *__restrict__ tmp,
                      double *__restrict__ dst)
    {
      const int       sx = g.ext_hi[0] - g.ext_lo[0] + 1, sy = g.ext_hi[1] - g.ext_lo[1] + 1;
      const int       m = f.m, m2 = m * m;
      const long long m3 = (long long)m2 * m, ncells = (long long)g.nc[0] * g.nc[1] * g.nc[2];
      // The DoFs of a z plane are taken in strips of H = 4 rows, x-major inside a strip (thread t of a strip: column t / 4,
      // row t % 4): the rows of a patch result (m doubles) that consecutive DoF rows read lie next to each other in tmp,
      // so the four rows of a strip share their cache lines inside one block; every XCD (own L2) takes a contiguous range
      // of blocks, i.e. a slab of planes (blocks b and b + 8 share an XCD).
      // (patches of fewer than 6 DoFs per direction: plain rows, measured faster there)
      const int hs = f.m >= 6 ? 2 : 0, H = 1 << hs;
      const int strip = H * sx, bpp = (((sy + H - 1) >> hs) * strip + BLOCK - 1) / BLOCK; // blocks per plane
      {
        const long long b = xcd_block(blockIdx.x, gridDim.x);
        const int       sk = (int)(b / bpp), e2 = (int)(b % bpp) * BLOCK + threadIdx.x;
        const int       t = e2 % strip, si = t >> hs, sj = H * (e2 / strip) + (t & (H - 1));
        if (si < sx && sj < sy)
        {
          GatherDir cx, cy, cz;
          gather_candidates(g, f, 0, si + g.ext_lo[0], cx);
          gather_candidates(g, f, 1, sj + g.ext_lo[1], cy);
          gather_candidates(g, f, 2, sk + g.ext_lo[2], cz);
          double sum = 0.;
          for (int kz = 0; kz < cz.n; ++kz)
            for (int ky = 0; ky < cy.n; ++ky)
              for (int kx = 0; kx < cx.n; ++kx)
                {
                  const long long cell = cx.cell[kx] + (long long)g.nc[0] * (cy.cell[ky] + (long long)g.nc[1] * cz.cell[kz]);
                  {
                    const long long local = cz.idx[kz] * m2 + cy.idx[ky] * m + cx.idx[kx];
                    // (the per-cell kernels store entry-major, tmp[entry][cell]: consecutive cells next to each other)
                    sum += f.small ? tmp[local * ncells + cell] : tmp[cell * m3 + local];
                  }
                }
          const long long off = si * g.stride[0] + sj * g.stride[1] + sk * g.stride[2];
          if (fu.mode != PC_PLAIN)
            {
              if (g.dirichlet(0, si + g.ext_lo[0]) || g.dirichlet(1, sj + g.ext_lo[1]) || g.dirichlet(2, sk + g.ext_lo[2]))
                sum = src[off]; // unit rows on constrained DoFs (include/matrix_free_internal.h:247-255)
              if (fu.mode == PC_FIRST)
                sum = fu.f2 * sum;
              else if (fu.mode == PC_STEP1)
                sum = fu.f1 * fu.x[off] + fu.f2 * sum;
              else
                {
                  const double xv = fu.x[off];
                  sum             = xv + fu.f1 * (xv - fu.x_old[off]) + fu.f2 * sum;
                }
            }
          dst[off] = sum;
        }
      }
    }

    // ---------------------------------------------------------------------
    // Laplace operator
    // ---------------------------------------------------------------------
    template <int P>
    struct OpCfg
    {
      static constexpr int N1  = P + 1;
      static constexpr int N2  = N1 * N1;
      static constexpr int N3  = N1 * N1 * N1;
      static constexpr int CPB = (BLOCK / N2) > 0 ? (BLOCK / N2) : 1;
    };

    // out[q] = sum_i A[q][i] in[i]
    template <int N, typename AT>
    __device__ __forceinline__ void
    apply_mat(const AT &A, double (&u)[N])
    {
      double t[N];
#pragma unroll
      for (int q = 0; q < N; ++q)
        {
          double s = 0.;
#pragma unroll
          for (int i = 0; i < N; ++i)
            s = fma(A[q * N + i], u[i], s);
          t[q] = s;
        }
#pragma unroll
      for (int q = 0; q < N; ++q)
        u[q] = t[q];
    }
    // out[i] = sum_q A[q][i] in[q]
    template <int N, typename AT>
    __device__ __forceinline__ void
    apply_mat_t(const AT &A, double (&u)[N])
    {
      double t[N];
#pragma unroll
      for (int i = 0; i < N; ++i)
        t[i] = 0.;
#pragma unroll
      for (int q = 0; q < N; ++q)
#pragma unroll
        for (int i = 0; i < N; ++i)
          t[i] = fma(A[q * N + i], u[q], t[i]);
#pragma unroll
      for (int i = 0; i < N; ++i)
        u[i] = t[i];
    }

    __device__ __forceinline__ double fast_reciprocal(const double x);

    // do_invert for 3x3, include/operator.h:891-914: t <- t^-1, returns det
    __device__ __forceinline__ double
    invert3(double (&t)[3][3])
    {
      const double tr00 = t[1][1] * t[2][2] - t[1][2] * t[2][1];
      const double tr10 = t[1][2] * t[2][0] - t[1][0] * t[2][2];
      const double tr20 = t[1][0] * t[2][1] - t[1][1] * t[2][0];
      const double det  = t[0][0] * tr00 + t[0][1] * tr10 + t[0][2] * tr20;
      const double inv  = fast_reciprocal(det);
      const double tr01 = t[0][2] * t[2][1] - t[0][1] * t[2][2];
      const double tr02 = t[0][1] * t[1][2] - t[0][2] * t[1][1];
      const double tr11 = t[0][0] * t[2][2] - t[0][2] * t[2][0];
      const double tr12 = t[0][2] * t[1][0] - t[0][0] * t[1][2];
      const double t21  = inv * (t[0][1] * t[2][0] - t[0][0] * t[2][1]);
      const double t22  = inv * (t[0][0] * t[1][1] - t[0][1] * t[1][0]);
      t[0][0]           = inv * tr00;
      t[0][1]           = inv * tr01;
      t[0][2]           = inv * tr02;
      t[1][0]           = inv * tr10;
      t[1][1]           = inv * tr11;
      t[1][2]           = inv * tr12;
      t[2][0]           = inv * tr20;
      t[2][1]           = t21;
      t[2][2]           = t22;
      return det;
    }

    // Jacobian rows jac[d][e] = dX_e/dxi_d of the tri-quadratic map given by its 27 monomial coefficient
    // vectors v[9k+3j+i][e], with the x and y powers contracted once per thread (the thread's quadrature
    // point (x, y) does not change along its z loop): 27 contracted coefficients
    //   cz[k][t][e] = sum_{i,j} w_t(i, j) v[9k+3j+i][e],  w_0 = dx_i py_j, w_1 = px_i dy_j, w_2 = px_i py_j,
    // then 9 + 9 + 6 fused multiply-adds per z instead of 243 (include/operator.h:1101-1128 collapses z, then y,
    // then x for the same reason).
    struct CollapsedJacobian
    {
      double cz[3][3][3];
      __device__ __forceinline__ void
      init(const double *__restrict__ v, const double x, const double y)
      {
        const double px[3] = {1., x, x * x}, py[3] = {1., y, y * y};
        const double dx[3] = {0., 1., 2. * x}, dy[3] = {0., 1., 2. * y};
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int e = 0; e < 3; ++e)
                cz[k][t][e] = 0.;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int i = 0; i < 3; ++i)
                {
                  const double w0 = dx[i] * py[j], w1 = px[i] * dy[j], w2 = px[i] * py[j];
#pragma unroll
                  for (int e = 0; e < 3; ++e)
                    {
                      const double c = v[(9 * k + 3 * j + i) * 3 + e];
                      if (i > 0)
                        cz[k][0][e] = fma(w0, c, cz[k][0][e]);
                      if (j > 0)
                        cz[k][1][e] = fma(w1, c, cz[k][1][e]);
                      cz[k][2][e] = fma(w2, c, cz[k][2][e]);
                    }
                }
          }
      }
      __device__ __forceinline__ void
      at(const double z, double (&jac)[3][3]) const
      {
        const double z2 = z * z, dz2 = 2. * z;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          {
            jac[0][e] = fma(z2, cz[2][0][e], fma(z, cz[1][0][e], cz[0][0][e]));
            jac[1][e] = fma(z2, cz[2][1][e], fma(z, cz[1][1][e], cz[0][1][e]));
            jac[2][e] = fma(dz2, cz[2][2][e], cz[1][2][e]);
          }
      }
    };

    // reciprocal with two Newton steps (full double precision, a fraction of an IEEE division)
    __device__ __forceinline__ double
    fast_reciprocal(const double x)
    {
      double r = __builtin_amdgcn_rcp(x);
      r        = fma(fma(-x, r, 1.0), r, r);
      r        = fma(fma(-x, r, 1.0), r, r);
      return r;
    }

    // DIAG = false: dst += A src (scatter-add);  DIAG = true: dst += diag(A)
    // CART: box cells (op.h) / deformed cells (op.coeff): one instantiation each, so that neither carries the
    // registers and the code of the other geometry
    template <int P, bool DIAG, bool CART>
    __global__ void __launch_bounds__(BLOCK, CART ? (P <= 5 ? 4 : 2) : (P <= 4 ? 3 : 2))
    laplace_generic_kernel(const DevGrid g, const DevOp op, const double *__restrict__ src, double *__restrict__ dst, const double sign)
    {
      using C           = OpCfg<P>;
      constexpr int N1 = C::N1, N2 = C::N2, N3 = C::N3;
      __shared__ double T0[C::CPB][N3], T1[C::CPB][N3], T2[C::CPB][N3];
      __shared__ double Gs[C::CPB][81];
      __shared__ double Ds[N2], xqs[N1], wqs[N1];
      // the interpolation matrix is the same for every cell: scalar operands instead of LDS reads
      const UniformMatrix Nu(op.N);

      const int  tid    = threadIdx.x;
      const int  lc     = tid / N2;
      const int  lt     = tid % N2;
      const int  ncells = g.nc[0] * g.nc[1] * g.nc[2];
      // (no xcd_block here: with neighbouring cells on one XCD the scatter's atomics collide more, +12 % measured)
      const int  cell   = blockIdx.x * C::CPB + lc;
      const bool active = (lc < C::CPB) && (cell < ncells);
      if (tid < N2)
        {
          Ds[tid] = op.Dc[tid];
        }
      if (tid < N1)
        {
          xqs[tid] = op.xq[tid];
          wqs[tid] = op.wq[tid];
        }
      int c3[3] = {0, 0, 0};
      if (active)
        {
          c3[0] = cell % g.nc[0];
          c3[1] = (cell / g.nc[0]) % g.nc[1];
          c3[2] = cell / (g.nc[0] * g.nc[1]);
          if (!CART)
            {
              const long long ce = (c3[0] + 1) + (g.nc[0] + 2) * ((c3[1] + 1) + (long long)(g.nc[1] + 2) * (c3[2] + 1));
              for (int e = lt; e < 81; e += N2)
                Gs[lc][e] = op.coeff[ce * 81 + e];
            }
        }
      const int a = lt % N1, b = lt / N1;

      double    u[N1], uin[N1];
      bool      valid[N1];
      long long off[N1];
      double    diag_acc[N1];
      {
        const int  I = P * c3[0] + a, J = P * c3[1] + b;
        const bool vxy = active && g.exists(0, I) && g.exists(1, J) && !g.dirichlet(0, I) && !g.dirichlet(1, J);
        const long long base = vxy ? (g.store_index(0, I) * g.stride[0] + g.store_index(1, J) * g.stride[1]) : 0;
#pragma unroll
        for (int k = 0; k < N1; ++k)
          {
            const int K = P * c3[2] + k;
            valid[k]    = vxy && g.exists(2, K) && !g.dirichlet(2, K);
            off[k]      = valid[k] ? base + g.store_index(2, K) * g.stride[2] : 0;
            uin[k]      = (!DIAG && valid[k]) ? src[off[k]] : 0.;
            diag_acc[k] = 0.;
          }
      }
      __syncthreads();

      for (int col = 0; col < (DIAG ? N3 : 1); ++col)
        {
          if (DIAG)
            {
#pragma unroll
              for (int k = 0; k < N1; ++k)
                uin[k] = (k * N2 + lt == col) ? 1. : 0.;
            }
#pragma unroll
          for (int k = 0; k < N1; ++k)
            u[k] = uin[k];
          // interpolate to quadrature points: z (registers), y, x
          if (active)
            {
              apply_mat<N1>(Nu, u);
#pragma unroll
              for (int k = 0; k < N1; ++k)
                T0[lc][k * N2 + b * N1 + a] = u[k];
            }
          __syncthreads();
          if (active)
            {
#pragma unroll
              for (int j = 0; j < N1; ++j)
                u[j] = T0[lc][b * N2 + j * N1 + a];
              apply_mat<N1>(Nu, u);
#pragma unroll
              for (int j = 0; j < N1; ++j)
                T0[lc][b * N2 + j * N1 + a] = u[j];
            }
          __syncthreads();
          if (active)
            {
#pragma unroll
              for (int i = 0; i < N1; ++i)
                u[i] = T0[lc][b * N2 + a * N1 + i];
              apply_mat<N1>(Nu, u);
#pragma unroll
              for (int i = 0; i < N1; ++i)
                T0[lc][b * N2 + a * N1 + i] = u[i];
            }
          __syncthreads();
          // gradients + geometry at the quadrature points: thread (qy=b, qx=a), loop qz
          double hz[N1];
          if (active)
            {
              double vz[N1];
#pragma unroll
              for (int q = 0; q < N1; ++q)
                vz[q] = T0[lc][q * N2 + b * N1 + a];
              double            hxc = 1., hyc = 1., hzc = 1.;
              CollapsedJacobian cj;
              if (CART)
                {
                  hxc = op.h[0][c3[0] + 1];
                  hyc = op.h[1][c3[1] + 1];
                  hzc = op.h[2][c3[2] + 1];
                }
              else
                cj.init(Gs[lc], xqs[a], xqs[b]);
#pragma unroll
              for (int qz = 0; qz < N1; ++qz)
                {
                  double gx = 0., gy = 0., gz = 0.;
#pragma unroll
                  for (int r = 0; r < N1; ++r)
                    {
                      gx = fma(Ds[a * N1 + r], T0[lc][qz * N2 + b * N1 + r], gx);
                      gy = fma(Ds[b * N1 + r], T0[lc][qz * N2 + r * N1 + a], gy);
                      gz = fma(Ds[qz * N1 + r], vz[r], gz);
                    }
                  const double w = wqs[a] * wqs[b] * wqs[qz];
                  double       ox, oy, oz;
                  if (CART)
                    {
                      const double det = hxc * hyc * hzc * w;
                      ox               = gx * det / (hxc * hxc);
                      oy               = gy * det / (hyc * hyc);
                      oz               = gz * det / (hzc * hzc);
                    }
                  else
                    {
                      double jac[3][3];
                      cj.at(xqs[qz], jac);
                      const double det = invert3(jac) * w;
                      // tmp = G grad * det ; out = G^T tmp   (operator.h:1132-1152)
                      const double t0 = (jac[0][0] * gx + jac[0][1] * gy + jac[0][2] * gz) * det;
                      const double t1 = (jac[1][0] * gx + jac[1][1] * gy + jac[1][2] * gz) * det;
                      const double t2 = (jac[2][0] * gx + jac[2][1] * gy + jac[2][2] * gz) * det;
                      ox              = jac[0][0] * t0 + jac[1][0] * t1 + jac[2][0] * t2;
                      oy              = jac[0][1] * t0 + jac[1][1] * t1 + jac[2][1] * t2;
                      oz              = jac[0][2] * t0 + jac[1][2] * t1 + jac[2][2] * t2;
                    }
                  T1[lc][qz * N2 + b * N1 + a] = ox;
                  T2[lc][qz * N2 + b * N1 + a] = oy;
                  hz[qz]                        = oz;
                }
            }
          __syncthreads();
          if (active)
            {
              // integrate: collocation-derivative transposed
#pragma unroll
              for (int qz = 0; qz < N1; ++qz)
                {
                  double s = 0.;
#pragma unroll
                  for (int r = 0; r < N1; ++r)
                    {
                      s = fma(Ds[r * N1 + a], T1[lc][qz * N2 + b * N1 + r], s);
                      s = fma(Ds[r * N1 + b], T2[lc][qz * N2 + r * N1 + a], s);
                      s = fma(Ds[r * N1 + qz], hz[r], s);
                    }
                  u[qz] = s;
                }
              apply_mat_t<N1>(Nu, u); // z
            }
          __syncthreads(); // all reads of T0 (vz, rows) are done
          if (active)
            {
#pragma unroll
              for (int k = 0; k < N1; ++k)
                T0[lc][k * N2 + b * N1 + a] = u[k];
            }
          __syncthreads();
          if (active)
            {
#pragma unroll
              for (int j = 0; j < N1; ++j)
                u[j] = T0[lc][b * N2 + j * N1 + a];
              apply_mat_t<N1>(Nu, u);
#pragma unroll
              for (int j = 0; j < N1; ++j)
                T0[lc][b * N2 + j * N1 + a] = u[j];
            }
          __syncthreads();
          if (active)
            {
#pragma unroll
              for (int i = 0; i < N1; ++i)
                u[i] = T0[lc][b * N2 + a * N1 + i];
              apply_mat_t<N1>(Nu, u);
#pragma unroll
              for (int i = 0; i < N1; ++i)
                T0[lc][b * N2 + a * N1 + i] = u[i];
            }
          __syncthreads();
          if (active)
            {
              // thread (j=b, i=a): column along z.  The cells of a block are consecutive in x: the last column of a cell
              // and the first one of the next cell of the block are the same DoFs - the right cell adds its neighbour's
              // column from LDS, the left one skips those atomics: a fifth fewer at degree 4, and no back-to-back updates of
              // one address by neighbouring threads, which serialise in the L2 (80^3 Cartesian cells: 0.94 -> 0.53 ms,
              // 78^3 deformed cells 0.86 -> 0.77 ms; a 5 x 2 tile of cells with x and y merges was slower again: 0.63 / 0.83)
              const bool take_left  = !DIAG && a == 0 && lc > 0 && c3[0] > 0;
              const bool give_right = !DIAG && a == P && lc + 1 < C::CPB && cell + 1 < ncells && c3[0] + 1 < g.nc[0];
#pragma unroll
              for (int k = 0; k < N1; ++k)
                {
                  double v = T0[lc][k * N2 + b * N1 + a];
                  if (take_left)
                    v += T0[lc - 1][k * N2 + b * N1 + P];
                  if (DIAG)
                    {
                      if (k * N2 + lt == col)
                        diag_acc[k] = v;
                    }
                  else if (valid[k] && !give_right)
                    unsafeAtomicAdd(&dst[off[k]], sign * v);
                }
            }
          __syncthreads();
        }
      if (DIAG && active)
        {
#pragma unroll
          for (int k = 0; k < N1; ++k)
            if (valid[k])
              unsafeAtomicAdd(&dst[off[k]], diag_acc[k]);
        }
    }
  } // namespace

  // -----------------------------------------------------------------------
  // launchers
  // -----------------------------------------------------------------------
  template <int M>
  static void
  launch_fdm(const DevGrid &g, const DevFdm &f, const double *src, double *dst, double *tmp, const FdmFused &fu, hipStream_t st)
  {
    const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
    const int blocks = (ncells + FdmCfg<M>::CPB - 1) / FdmCfg<M>::CPB;
    if (tmp == nullptr && fu.direct == 2)
      {
        // the cells outside the slice-per-lane box, added with atomics (fdm_slice7_acc)
        if constexpr (M == 7)
          {
            const int others = (f.s7_n_others + FdmCfg<M>::CPB - 1) / FdmCfg<M>::CPB;
            if (others > 0)
              hipLaunchKernelGGL((fdm_generic_kernel<M, false>), dim3(others), dim3(BLOCK), 0, st, g, f, fu, src, dst);
          }
        return;
      }
    if (tmp == nullptr)
      {
        if (fu.direct || !(M <= 3 && f.small && fdm_cell(g, f, src, dst, false, st)))
          hipLaunchKernelGGL((fdm_generic_kernel<M, false>), dim3(blocks), dim3(BLOCK), 0, st, g, f, fu, src, dst);
        return;
      }
    if (M <= 3 && f.small && fdm_cell(g, f, src, tmp, true, st))
      ;
    else if (M == 7 && f.s7_on)
      {
        const int others = (f.s7_n_others + FdmCfg<M>::CPB - 1) / FdmCfg<M>::CPB;
        if (others > 0)
          hipLaunchKernelGGL((fdm_generic_kernel<M, true>), dim3(others), dim3(BLOCK), 0, st, g, f, fu, src, tmp);
        fdm_slice7(g, f, src, tmp, st);
      }
    else
      hipLaunchKernelGGL((fdm_generic_kernel<M, true>), dim3(blocks), dim3(BLOCK), 0, st, g, f, fu, src, tmp);
    const int       sx = g.ext_hi[0] - g.ext_lo[0] + 1, sy = g.ext_hi[1] - g.ext_lo[1] + 1;
    const int       H  = f.m >= 6 ? 4 : 1; // rows per strip, as in the kernel
    const long long gb = (long long)((((sy + H - 1) / H) * H * sx + BLOCK - 1) / BLOCK) * (g.ext_hi[2] - g.ext_lo[2] + 1);
    hipLaunchKernelGGL(fdm_gather_kernel, dim3((unsigned)gb), dim3(BLOCK), 0, st, g, f, fu, src, tmp, dst);
  }

  bool
  fdm_generic(const DevGrid &g, const DevFdm &f, const double *src, double *dst, double *tmp, const FdmFused &fu, hipStream_t st)
  {
    switch (f.m)
      {
        case 2: launch_fdm<2>(g, f, src, dst, tmp, fu, st); break;
        case 3: launch_fdm<3>(g, f, src, dst, tmp, fu, st); break;
        case 4: launch_fdm<4>(g, f, src, dst, tmp, fu, st); break;
        case 5: launch_fdm<5>(g, f, src, dst, tmp, fu, st); break;
        case 6: launch_fdm<6>(g, f, src, dst, tmp, fu, st); break;
        case 7: launch_fdm<7>(g, f, src, dst, tmp, fu, st); break;
        case 8: launch_fdm<8>(g, f, src, dst, tmp, fu, st); break;
        case 9: launch_fdm<9>(g, f, src, dst, tmp, fu, st); break;
        case 10: launch_fdm<10>(g, f, src, dst, tmp, fu, st); break;
        case 11: launch_fdm<11>(g, f, src, dst, tmp, fu, st); break;
        case 12: launch_fdm<12>(g, f, src, dst, tmp, fu, st); break;
        case 13: launch_fdm<13>(g, f, src, dst, tmp, fu, st); break;
        default: return false;
      }
    return true;
  }

  template <int P, bool DIAG>
  static void
  launch_op(const DevGrid &g, const DevOp &op, const double *src, double *dst, hipStream_t st, const double sign)
  {
    const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
    const int blocks = (ncells + OpCfg<P>::CPB - 1) / OpCfg<P>::CPB;
    if (op.cartesian)
      hipLaunchKernelGGL((laplace_generic_kernel<P, DIAG, true>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
    else
      hipLaunchKernelGGL((laplace_generic_kernel<P, DIAG, false>), dim3(blocks), dim3(BLOCK), 0, st, g, op, src, dst, sign);
  }

  bool
  laplace_generic(const DevGrid &g, const DevOp &op, const double *src, double *dst, bool diag, hipStream_t st, const double sign)
  {
    if (op.small && laplace_cell(g, op, src, dst, diag, st, sign))
      return true;
#define ASM_CASE(P)                              \
  case P:                                        \
    if (diag)                                    \
      launch_op<P, true>(g, op, src, dst, st, sign);   \
    else                                         \
      launch_op<P, false>(g, op, src, dst, st, sign);  \
    break;
    switch (g.p)
      {
        ASM_CASE(1)
        ASM_CASE(2)
        ASM_CASE(3)
        ASM_CASE(4)
        ASM_CASE(5)
        ASM_CASE(6)
        ASM_CASE(7)
        default: return false;
      }
#undef ASM_CASE
    return true;
  }
} // namespace asmamd
