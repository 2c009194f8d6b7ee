// Prototype: thread-per-cell FDM (5x5x5) entirely in registers.  Measures FMA throughput,
// register allocation and code size of the fully unrolled 6-sweep core.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int M = 5;

template <int STRIDE, bool FWD>
__device__ __forceinline__ void sweep_line(double *u, const int base, const double *Q)
{
  double in[M], out[M];
#pragma unroll
  for (int i = 0; i < M; ++i) in[i] = u[base + i * STRIDE];
#pragma unroll
  for (int a = 0; a < M; ++a)
    {
      double s = 0.;
#pragma unroll
      for (int i = 0; i < M; ++i) s = fma(FWD ? Q[i * M + a] : Q[a * M + i], in[i], s);
      out[a] = s;
    }
#pragma unroll
  for (int i = 0; i < M; ++i) u[base + i * STRIDE] = out[i];
}

template <bool PERLANE>
__global__ void __launch_bounds__(256, 1) cell_kernel(double *out, const double *in, const double *qtab, int iters)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  double u[125];
#pragma unroll
  for (int e = 0; e < 125; ++e) u[e] = in[(size_t)e * gridDim.x * 256 + t];
  double Q[25];
  for (int it = 0; it < iters; ++it)
    {
      const double *q = qtab + (PERLANE ? (threadIdx.x & 3) * 90 : 0);
#pragma unroll
      for (int dir = 0; dir < 3; ++dir)
        {
#pragma unroll
          for (int e = 0; e < 25; ++e) Q[e] = q[dir * 30 + e];
#pragma unroll
          for (int l = 0; l < 25; ++l)
            {
              if (dir == 0) sweep_line<1, true>(u, l * 5, Q);
              else if (dir == 1) sweep_line<5, true>(u, (l / 5) * 25 + (l % 5), Q);
              else sweep_line<25, true>(u, l, Q);
            }
        }
#pragma unroll
      for (int e = 0; e < 125; ++e) u[e] *= q[25 + (e % 5)] ;
#pragma unroll
      for (int dir = 2; dir >= 0; --dir)
        {
#pragma unroll
          for (int e = 0; e < 25; ++e) Q[e] = q[dir * 30 + e];
#pragma unroll
          for (int l = 0; l < 25; ++l)
            {
              if (dir == 0) sweep_line<1, false>(u, l * 5, Q);
              else if (dir == 1) sweep_line<5, false>(u, (l / 5) * 25 + (l % 5), Q);
              else sweep_line<25, false>(u, l, Q);
            }
        }
    }
#pragma unroll
  for (int e = 0; e < 125; ++e) out[(size_t)e * gridDim.x * 256 + t] = u[e];
}

int main()
{
  const int blocks = 256 * 2, iters = 20;
  const size_t n = (size_t)blocks * 256 * 125;
  double *in, *out, *q;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMalloc(&q, 4 * 90 * 8);
  std::vector<double> h(n, 0.5), hq(360, 0.2);
  hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(q, hq.data(), 360 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep)
        {
          hipEventRecord(e0);
          if (mode == 0) cell_kernel<false><<<blocks, 256>>>(out, in, q, iters); else cell_kernel<true><<<blocks, 256>>>(out, in, q, iters);
          hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
      const double fma = (double)blocks * 256 * iters * 3750;
      printf("thread-per-cell %s: %.3f ms  %.2f TFMA/s  %.2f Gcells/s\n", mode ? "per-lane Q" : "uniform Q", ms, fma / ms / 1e9,
             (double)blocks * 256 * iters / ms / 1e6);
    }
  return 0;
}
