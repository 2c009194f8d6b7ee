// Streaming vector kernels: Chebyshev updates (the "pre/post operations" of
// PreconditionChebyshev that the reference passes as lambdas into
// vmult(dst, src, pre, post), include/matrix_free.h:960-986), reductions for the
// eigenvalue estimate, constrained-DoF handling
// (MFWorker::apply_operation_to_constrained_dofs, include/matrix_free_internal.h:247-255)
// and box pack/unpack for ghost exchange (VectorDataExchange, :3-109).
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace asmamd
{
  namespace
  {
    constexpr int VB = 256;

    inline int
    grid_for(const long long n, const int per_thread)
    {
      const long long blocks = (n + (long long)VB * per_thread - 1) / ((long long)VB * per_thread);
      return (int)std::min<long long>(std::max<long long>(blocks, 1), 256 * 16);
    }

    template <typename F>
    __global__ void __launch_bounds__(VB) elementwise2_kernel(const long long n2, F f)
    {
      // two doubles per lane and iteration (16-byte accesses)
      for (long long i = blockIdx.x * (long long)VB + threadIdx.x; i < n2; i += (long long)gridDim.x * VB)
        f(i);
    }

    struct ResidualOp
    {
      double2 *      r;
      const double2 *b;
      __device__ void operator()(const long long i) const
      {
        const double2 bv = b[i];
        double2       rv = r[i];
        rv.x             = bv.x - rv.x;
        rv.y             = bv.y - rv.y;
        r[i]             = rv;
      }
    };
    struct ScaleOp
    {
      double2 *      y;
      const double2 *x;
      double         a;
      __device__ void operator()(const long long i) const
      {
        double2 v = x[i];
        v.x *= a;
        v.y *= a;
        y[i] = v;
      }
    };
    struct MulOp
    {
      double2 *      y;
      const double2 *x, *d;
      __device__ void operator()(const long long i) const
      {
        double2       v = x[i];
        const double2 w = d[i];
        v.x *= w.x;
        v.y *= w.y;
        y[i] = v;
      }
    };
    struct AxpbyOp
    {
      double2 *      y;
      const double2 *x;
      double         a, b;
      __device__ void operator()(const long long i) const
      {
        const double2 xv = x[i];
        double2       yv = y[i];
        yv.x             = a * xv.x + b * yv.x;
        yv.y             = a * xv.y + b * yv.y;
        y[i]             = yv;
      }
    };
    struct ChebOp
    {
      double2 *      y;
      const double2 *x, *xo, *t;
      double         f1, f2;
      __device__ void operator()(const long long i) const
      {
        const double2 xv = x[i], ov = xo[i], tv = t[i];
        double2       r;
        r.x  = xv.x + f1 * (xv.x - ov.x) + f2 * tv.x;
        r.y  = xv.y + f1 * (xv.y - ov.y) + f2 * tv.y;
        y[i] = r;
      }
    };
    struct InvDiagOp
    {
      double2 *d;
      __device__ void operator()(const long long i) const
      {
        double2 v = d[i];
        v.x       = fabs(v.x) > 1.0e-10 ? 1.0 / v.x : 1.0; // operator.h:1522-1523
        v.y       = fabs(v.y) > 1.0e-10 ? 1.0 / v.y : 1.0;
        d[i]      = v;
      }
    };

    template <typename F>
    void
    launch2(const long long n, F f, hipStream_t st)
    {
      // allocations are multiples of 8 doubles (setup_layout pads the row pitch)
      const long long n2 = n / 2;
      hipLaunchKernelGGL(elementwise2_kernel<F>, dim3(grid_for(n2, 4)), dim3(VB), 0, st, n2, f);
    }

    __device__ __forceinline__ long long
    owned_offset(const DevGrid &g, const long long e, int &I, int &J, int &K)
    {
      const int nx = g.own_hi[0] + 1, ny = g.own_hi[1] + 1;
      I            = (int)(e % nx);
      J            = (int)((e / nx) % ny);
      K            = (int)(e / ((long long)nx * ny));
      return g.offset(I, J, K);
    }

    __global__ void __launch_bounds__(VB)
    dot_kernel(const DevGrid g, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ result)
    {
      const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
      double          s = 0.;
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int             I, J, K;
          const long long o = owned_offset(g, e, I, J, K);
          s                 = fma(x[o], y[o], s);
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
      __shared__ double part[VB / 64];
      if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          double t = 0.;
          for (int w = 0; w < VB / 64; ++w)
            t += part[w];
          unsafeAtomicAdd(result, t);
        }
    }

    // result[i] += <w, L.v[i]> over the owned box, i < L.n <= 8: w is read once for all of them (the inner products of
    // one Gram-Schmidt step)
    __global__ void __launch_bounds__(VB)
    multi_dot_kernel(const DevGrid g, const double *__restrict__ w, const VecList L, double *__restrict__ result)
    {
      const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
      double          s[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int             I, J, K;
          const long long o  = owned_offset(g, e, I, J, K);
          const double    wv = w[o];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < L.n)
              s[i] = fma(wv, L.v[i][o], s[i]);
        }
      __shared__ double part[8][VB / 64];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        {
          double t = s[i];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1)
            t += __shfl_down(t, off, 64);
          if ((threadIdx.x & 63) == 0)
            part[i][threadIdx.x >> 6] = t;
        }
      __syncthreads();
      if (threadIdx.x < L.n)
        {
          double t = 0.;
          for (int q = 0; q < VB / 64; ++q)
            t += part[threadIdx.x][q];
          unsafeAtomicAdd(result + threadIdx.x, t);
        }
    }

    // w += sum_i L.a[i] L.v[i] over the whole allocation; result (if given) += |w_new|^2 over the owned box
    __global__ void __launch_bounds__(VB)
    multi_axpy_kernel(const DevGrid g, double *__restrict__ w, const VecList L, const long long n_alloc, double *__restrict__ result)
    {
      double acc = 0.;
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n_alloc; e += (long long)gridDim.x * VB)
        {
          double t = w[e];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < L.n)
              t = fma(L.a[i], L.v[i][e], t);
          w[e] = t;
          if (result != nullptr)
            {
              const long long K = e / g.stride[2], r = e - K * g.stride[2];
              const long long J = r / g.stride[1], I = r - J * g.stride[1];
              const bool own = I + g.ext_lo[0] >= 0 && I + g.ext_lo[0] <= g.own_hi[0] && J + g.ext_lo[1] >= 0 && J + g.ext_lo[1] <= g.own_hi[1] &&
                               K + g.ext_lo[2] >= 0 && K + g.ext_lo[2] <= g.own_hi[2];
              if (own)
                acc = fma(t, t, acc);
            }
        }
      if (result == nullptr)
        return;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
      __shared__ double part[VB / 64];
      if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          double t = 0.;
          for (int q = 0; q < VB / 64; ++q)
            t += part[q];
          unsafeAtomicAdd(result, t);
        }
    }

    __global__ void __launch_bounds__(VB)
    initial_guess_kernel(const DevGrid g, double *__restrict__ v, const long long gb0, const long long gb1, const long long gb2,
                         const long long gs0, const long long gs1)
    {
      const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int             I, J, K;
          const long long o   = owned_offset(g, e, I, J, K);
          const long long gid = (gb0 + I) + gs0 * ((gb1 + J) + gs1 * (gb2 + K));
          v[o]                = (double)(gid % 11);
        }
    }

    __global__ void __launch_bounds__(VB) add_scalar_owned_kernel(const DevGrid g, double *__restrict__ v, const double a)
    {
      const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int             I, J, K;
          const long long o = owned_offset(g, e, I, J, K);
          v[o] += a;
        }
    }

    // one launch over the (up to) six Dirichlet planes of the block
    __global__ void __launch_bounds__(VB)
    constrained_kernel(const DevGrid g, const int mode, double *__restrict__ dst, const double *__restrict__ src,
                       const double *__restrict__ b)
    {
      const int face = blockIdx.y; // 2*d + side
      const int d = face / 2, side = face % 2;
      if (!(side ? g.dir_hi[d] : g.dir_lo[d]))
        return;
      const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      // all stored planes of the two tangential directions
      const int lo1 = g.wrap[d1] ? 0 : g.ext_lo[d1], n1 = (g.wrap[d1] ? g.nc[d1] * g.p - 1 : g.ext_hi[d1]) - lo1 + 1;
      const int lo2 = g.wrap[d2] ? 0 : g.ext_lo[d2], n2 = (g.wrap[d2] ? g.nc[d2] * g.p - 1 : g.ext_hi[d2]) - lo2 + 1;
      const long long n = (long long)n1 * n2;
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int c[3];
          c[d]  = side ? g.nc[d] * g.p : 0;
          c[d1] = lo1 + (int)(e % n1);
          c[d2] = lo2 + (int)(e / n1);
          const long long o = g.offset(c[0], c[1], c[2]);
          dst[o]            = mode == 0 ? 0. : (mode == 1 ? src[o] : b[o] - src[o]);
        }
    }

    __global__ void __launch_bounds__(VB)
    constrained_fused_kernel(const DevGrid g, const FdmFused fu, double *__restrict__ dst, const double *__restrict__ src)
    {
      const int face = blockIdx.y; // 2*d + side
      const int d = face / 2, side = face % 2;
      if (!(side ? g.dir_hi[d] : g.dir_lo[d]))
        return;
      const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      const int lo1 = g.wrap[d1] ? 0 : g.ext_lo[d1], n1 = (g.wrap[d1] ? g.nc[d1] * g.p - 1 : g.ext_hi[d1]) - lo1 + 1;
      const int lo2 = g.wrap[d2] ? 0 : g.ext_lo[d2], n2 = (g.wrap[d2] ? g.nc[d2] * g.p - 1 : g.ext_hi[d2]) - lo2 + 1;
      const long long n = (long long)n1 * n2;
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          int c[3];
          c[d]  = side ? g.nc[d] * g.p : 0;
          c[d1] = lo1 + (int)(e % n1);
          c[d2] = lo2 + (int)(e / n1);
          // a DoF on an edge or a corner lies on two or three of the planes: the update is not idempotent when dst is
          // x_old, so only the plane of the lowest direction takes it
          bool earlier = false;
          for (int dd = 0; dd < d; ++dd)
            earlier = earlier || g.dirichlet(dd, c[dd]);
          if (earlier)
            continue;
          const long long o = g.offset(c[0], c[1], c[2]);
          double          t = src[o];
          if (fu.mode == PC_PLAIN)
            t = 0.;
          else if (fu.mode == PC_FIRST)
            t = fu.f2 * t;
          else if (fu.mode == PC_STEP1)
            t = fu.f1 * fu.x[o] + fu.f2 * t;
          else
            {
              const double xv = fu.x[o];
              t               = xv + fu.f1 * (xv - fu.x_old[o]) + fu.f2 * t;
            }
          dst[o] = t;
        }
    }

    __global__ void __launch_bounds__(VB)
    box_kernel(const DevGrid g, const int mode, double *__restrict__ vec, double *__restrict__ buf, const int l0, const int l1,
               const int l2, const int s0, const int s1, const int s2)
    {
      const long long n = (long long)s0 * s1 * s2;
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          const int       i = (int)(e % s0), j = (int)((e / s0) % s1), k = (int)(e / ((long long)s0 * s1));
          const long long o = g.offset(l0 + i, l1 + j, l2 + k);
          if (mode == 0)
            buf[e] = vec[o];
          else if (mode == 1)
            vec[o] = buf[e];
          else if (mode == 2)
            vec[o] += buf[e];
          else
            vec[o] = 0.;
        }
    }
    __global__ void __launch_bounds__(VB)
    boxes_kernel(const DevGrid g, const int mode, double *__restrict__ vec, double *__restrict__ buf, const BoxList bl)
    {
      const int       b  = blockIdx.y;
      const int       s0 = bl.size[b][0], s1 = bl.size[b][1], s2 = bl.size[b][2];
      const long long n  = (long long)s0 * s1 * s2;
      double *        bb = buf + (mode == 3 ? 0 : bl.offset[b]);
      for (long long e = blockIdx.x * (long long)VB + threadIdx.x; e < n; e += (long long)gridDim.x * VB)
        {
          const int       i = (int)(e % s0), j = (int)((e / s0) % s1), k = (int)(e / ((long long)s0 * s1));
          const long long o = g.offset(bl.lo[b][0] + i, bl.lo[b][1] + j, bl.lo[b][2] + k);
          if (mode == 0)
            bb[e] = vec[o];
          else if (mode == 1)
            vec[o] = bb[e];
          else if (mode == 2)
            unsafeAtomicAdd(&vec[o], bb[e]); // the boxes of a compress(add) overlap in the owned region (faces, edges, corners)
          else
            vec[o] = 0.;
        }
    }
  } // namespace

  void
  vec_residual(double *r, const double *b, long long n, hipStream_t st)
  {
    launch2(n, ResidualOp{(double2 *)r, (const double2 *)b}, st);
  }
  void
  vec_scale(double *y, const double *x, double a, long long n, hipStream_t st)
  {
    launch2(n, ScaleOp{(double2 *)y, (const double2 *)x, a}, st);
  }
  void
  vec_mul(double *y, const double *x, const double *d, long long n, hipStream_t st)
  {
    launch2(n, MulOp{(double2 *)y, (const double2 *)x, (const double2 *)d}, st);
  }
  void
  vec_axpby(double *y, double a, const double *x, double b, long long n, hipStream_t st)
  {
    launch2(n, AxpbyOp{(double2 *)y, (const double2 *)x, a, b}, st);
  }
  void
  vec_cheb_update(double *y, const double *x, const double *xo, const double *t, double f1, double f2, long long n,
                  hipStream_t st)
  {
    launch2(n, ChebOp{(double2 *)y, (const double2 *)x, (const double2 *)xo, (const double2 *)t, f1, f2}, st);
  }
  void
  vec_invert_diagonal(double *d, long long n, hipStream_t st)
  {
    launch2(n, InvDiagOp{(double2 *)d}, st);
  }
  void
  vec_dot(const DevGrid &g, const double *x, const double *y, double *result, hipStream_t st)
  {
    const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
    hipLaunchKernelGGL(dot_kernel, dim3(grid_for(n, 8)), dim3(VB), 0, st, g, x, y, result);
  }
  void
  vec_multi_dot(const DevGrid &g, const double *w, const VecList &L, double *result, hipStream_t st)
  {
    const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
    hipLaunchKernelGGL(multi_dot_kernel, dim3(grid_for(n, 8)), dim3(VB), 0, st, g, w, L, result);
  }
  void
  vec_multi_axpy(const DevGrid &g, double *w, const VecList &L, long long n_alloc, double *result, hipStream_t st)
  {
    hipLaunchKernelGGL(multi_axpy_kernel, dim3(grid_for(n_alloc, 8)), dim3(VB), 0, st, g, w, L, n_alloc, result);
  }
  void
  vec_initial_guess(const DevGrid &g, double *v, const long long gbegin[3], const long long gsize[3], hipStream_t st)
  {
    const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
    hipLaunchKernelGGL(initial_guess_kernel, dim3(grid_for(n, 4)), dim3(VB), 0, st, g, v, gbegin[0], gbegin[1], gbegin[2],
                       gsize[0], gsize[1]);
  }
  void
  vec_add_scalar_owned(const DevGrid &g, double *v, double a, hipStream_t st)
  {
    const long long n = (long long)(g.own_hi[0] + 1) * (g.own_hi[1] + 1) * (g.own_hi[2] + 1);
    hipLaunchKernelGGL(add_scalar_owned_kernel, dim3(grid_for(n, 4)), dim3(VB), 0, st, g, v, a);
  }
  namespace
  {
    // blocks per face of the kernels over the Dirichlet planes: one thread per DoF of the largest face (64 blocks per
    // face, as these kernels were launched until round 4, left every thread six dependent round trips on the 80^3 block:
    // 20 us behind every sweep of the wave-pair kernels, which leave the constrained DoFs to them)
    unsigned
    face_blocks(const DevGrid &g, const int threads)
    {
      long long n = 1;
      for (int d = 0; d < 3; ++d)
        {
          const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
          const long long n1 = g.wrap[d1] ? g.nc[d1] * g.p : g.ext_hi[d1] - g.ext_lo[d1] + 1, n2 = g.wrap[d2] ? g.nc[d2] * g.p : g.ext_hi[d2] - g.ext_lo[d2] + 1;
          n = std::max(n, n1 * n2);
        }
      return (unsigned)std::max<long long>(1, std::min<long long>((n + threads - 1) / threads, 2048));
    }
  } // namespace
  void
  vec_constrained(const DevGrid &g, int mode, double *dst, const double *src, const double *b, hipStream_t st)
  {
    bool any = false;
    for (int d = 0; d < 3; ++d)
      any = any || g.dir_lo[d] || g.dir_hi[d];
    if (!any)
      return;
    hipLaunchKernelGGL(constrained_kernel, dim3(face_blocks(g, VB), 6), dim3(VB), 0, st, g, mode, dst, src, b);
  }
  namespace
  {
    // dst = b, on Dirichlet planes b - x: what a residual sweep that adds -A x to dst starts from (one launch instead of a
    // copy and a pass over the Dirichlet planes; the sweeps neither read nor write constrained rows)
    __global__ void __launch_bounds__(VB)
    residual_init_kernel(const DevGrid g, double *__restrict__ dst, const double *__restrict__ b, const double *__restrict__ x)
    {
      const unsigned sx = g.wrap[0] ? g.nc[0] * g.p : g.ext_hi[0] - g.ext_lo[0] + 1, sy = g.wrap[1] ? g.nc[1] * g.p : g.ext_hi[1] - g.ext_lo[1] + 1,
                     sz = g.wrap[2] ? g.nc[2] * g.p : g.ext_hi[2] - g.ext_lo[2] + 1;
      const unsigned n = sx * sy * sz;
      for (unsigned e = blockIdx.x * VB + threadIdx.x; e < n; e += gridDim.x * VB)
        {
          const unsigned k = e / (sx * sy), r = e - k * (sx * sy), j = r / sx, i = r - j * sx;
          const long long o = i * g.stride[0] + j * g.stride[1] + k * g.stride[2];
          const bool      c = g.dirichlet(0, (int)i + (g.wrap[0] ? 0 : g.ext_lo[0])) || g.dirichlet(1, (int)j + (g.wrap[1] ? 0 : g.ext_lo[1])) ||
                         g.dirichlet(2, (int)k + (g.wrap[2] ? 0 : g.ext_lo[2]));
          dst[o] = c ? b[o] - x[o] : b[o];
        }
    }
  } // namespace
  void
  vec_residual_init(const DevGrid &g, double *dst, const double *b, const double *x, hipStream_t st)
  {
    const long long sx = g.wrap[0] ? g.nc[0] * g.p : g.ext_hi[0] - g.ext_lo[0] + 1, sy = g.wrap[1] ? g.nc[1] * g.p : g.ext_hi[1] - g.ext_lo[1] + 1,
                    sz = g.wrap[2] ? g.nc[2] * g.p : g.ext_hi[2] - g.ext_lo[2] + 1;
    hipLaunchKernelGGL(residual_init_kernel, dim3(grid_for(sx * sy * sz, 4)), dim3(VB), 0, st, g, dst, b, x);
  }
  void
  vec_constrained_fused(const DevGrid &g, const FdmFused &fu, double *dst, const double *src, hipStream_t st)
  {
    bool any = false;
    for (int d = 0; d < 3; ++d)
      any = any || g.dir_lo[d] || g.dir_hi[d];
    if (!any)
      return;
    hipLaunchKernelGGL(constrained_fused_kernel, dim3(face_blocks(g, VB), 6), dim3(VB), 0, st, g, fu, dst, src);
  }
  namespace
  {
    // a box copy the kernel can execute: a known mode, the operands it dereferences, a box inside the stored extent
    bool
    box_valid(const DevGrid &g, const int mode, const double *vec, const double *buf, const int lo[3], const int size[3])
    {
      if (mode < 0 || mode > 3 || vec == nullptr || (mode != 3 && buf == nullptr))
        return false;
      for (int d = 0; d < 3; ++d)
        {
          const int n_store = g.wrap[d] ? g.nc[d] * g.p : g.ext_hi[d] - g.ext_lo[d] + 1;
          if (size[d] < 0 || size[d] > n_store || (!g.wrap[d] && (lo[d] < g.ext_lo[d] || lo[d] + size[d] - 1 > g.ext_hi[d])))
            return false;
        }
      return true;
    }
  } // namespace
  bool
  vec_box(const DevGrid &g, int mode, double *vec, double *buf, const int lo[3], const int size[3], hipStream_t st)
  {
    if (!box_valid(g, mode, vec, buf, lo, size))
      return false;
    const long long n = (long long)size[0] * size[1] * size[2];
    if (n <= 0)
      return true;
    hipLaunchKernelGGL(box_kernel, dim3(grid_for(n, 4)), dim3(VB), 0, st, g, mode, vec, buf, lo[0], lo[1], lo[2], size[0],
                       size[1], size[2]);
    return true;
  }
  bool
  vec_boxes(const DevGrid &g, int mode, double *vec, double *buf, const BoxList &boxes, hipStream_t st)
  {
    if (boxes.n < 0 || boxes.n > 26)
      return false;
    for (int b = 0; b < boxes.n; ++b)
      if (!box_valid(g, mode, vec, buf, boxes.lo[b], boxes.size[b]) || boxes.offset[b] < 0)
        return false;
    if (boxes.n == 0)
      return true;
    long long largest = 1;
    for (int b = 0; b < boxes.n; ++b)
      largest = std::max(largest, (long long)boxes.size[b][0] * boxes.size[b][1] * boxes.size[b][2]);
    hipLaunchKernelGGL(boxes_kernel, dim3(grid_for(largest, 4), boxes.n), dim3(VB), 0, st, g, mode, vec, buf, boxes);
    return true;
  }
  namespace
  {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __global__ void __launch_bounds__(256)
    stream_copy_kernel(v4f *__restrict__ dst, const v4f *__restrict__ src, const long long n)
    {
      const long long i = blockIdx.x * 1024ll + threadIdx.x;
      v4f             v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 256ll * u < n)
          v[u] = __builtin_nontemporal_load(src + i + 256ll * u);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 256ll * u < n)
          __builtin_nontemporal_store(v[u], dst + i + 256ll * u);
    }
  } // namespace

  void
  vec_stream_copy(void *dst, const void *src, const long long n_vec16, hipStream_t st)
  {
    const long long blocks = (n_vec16 + 1023) / 1024;
    if (blocks > 0 && blocks < (1ll << 31))
      hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (v4f *)dst, (const v4f *)src, n_vec16);
  }
} // namespace asmamd
