// Micro-benchmarks that size the tiled kernels: fp64 FMA issue rate per SIMD at
// 1..8 waves, with VGPR / SGPR / LDS-broadcast operands; plain stream copy rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) fma_kernel(double *out, const double *q, int iters)
{
  __shared__ double sq[32];
  if (threadIdx.x < 25) sq[threadIdx.x] = q[threadIdx.x];
  __syncthreads();
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4;
  double u0 = 0.1 * threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4;
  double c[25];
  for (int i = 0; i < 25; ++i) c[i] = q[i];  // uniform -> SGPR candidates
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int r = 0; r < 5; ++r)
        {
          double m0, m1, m2, m3, m4;
          if (MODE == 0) { m0 = c[r*5+0]; m1 = c[r*5+1]; m2 = c[r*5+2]; m3 = c[r*5+3]; m4 = c[r*5+4]; }
          else { m0 = sq[r*5+0]; m1 = sq[r*5+1]; m2 = sq[r*5+2]; m3 = sq[r*5+3]; m4 = sq[r*5+4]; }
          a0 = fma(m0, u0, a0); a1 = fma(m1, u0, a1); a2 = fma(m2, u0, a2); a3 = fma(m3, u0, a3); a4 = fma(m4, u0, a4);
          a0 = fma(m1, u1, a0); a1 = fma(m2, u1, a1); a2 = fma(m3, u1, a2); a3 = fma(m4, u1, a3); a4 = fma(m0, u1, a4);
          a0 = fma(m2, u2, a0); a1 = fma(m3, u2, a1); a2 = fma(m4, u2, a2); a3 = fma(m0, u2, a3); a4 = fma(m1, u2, a4);
          a0 = fma(m3, u3, a0); a1 = fma(m4, u3, a1); a2 = fma(m0, u3, a2); a3 = fma(m1, u3, a3); a4 = fma(m2, u3, a4);
          a0 = fma(m4, u4, a0); a1 = fma(m0, u4, a1); a2 = fma(m1, u4, a2); a3 = fma(m2, u4, a3); a4 = fma(m3, u4, a4);
          u0 = a0 * 1e-3; u1 = a1 * 1e-3; u2 = a2 * 1e-3; u3 = a3 * 1e-3; u4 = a4 * 1e-3;
        }
    }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4;
}

__global__ void __launch_bounds__(256) copy_kernel(double2 *dst, const double2 *src, long long n)
{
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) dst[i] = src[i];
}
// four 16-byte loads in flight per lane before the first store, streaming (non-temporal) accesses
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) copy4_kernel(v2d *dst, const v2d *src, long long n)
{
  const long long stride = gridDim.x * 256ll;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i + 3 * stride < n; i += 4 * stride)
    {
      const v2d v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride),
                    v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
      __builtin_nontemporal_store(v0, dst + i);
      __builtin_nontemporal_store(v1, dst + i + stride);
      __builtin_nontemporal_store(v2, dst + i + 2 * stride);
      __builtin_nontemporal_store(v3, dst + i + 3 * stride);
    }
}
__global__ void __launch_bounds__(256) triad_kernel(double2 *dst, const double2 *a, const double2 *b, long long n)
{
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    { double2 x = a[i], y = b[i]; x.x += 2.0 * y.x; x.y += 2.0 * y.y; dst[i] = x; }
}

int main()
{
  double *out, *q; hipMalloc(&out, 256 * 256 * 64 * 8); hipMalloc(&q, 32 * 8);
  std::vector<double> hq(32, 0.999); hipMemcpy(q, hq.data(), 32 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int bpc : {1, 2, 4, 8})  // blocks of 256 threads per CU -> waves per SIMD
      {
        const int blocks = 256 * bpc;
        for (int rep = 0; rep < 2; ++rep)
          {
            hipEventRecord(e0);
            if (mode == 0) fma_kernel<0><<<blocks, 256>>>(out, q, iters); else fma_kernel<1><<<blocks, 256>>>(out, q, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
          }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fmas = (double)blocks * 256 * iters * 5 * 25;  // + 25 muls per iter not counted
        printf("fma mode=%s waves/SIMD=%d : %.1f TFMA/s (%.1f TFLOP/s)  %.3f ms\n", mode ? "LDS-operand" : "reg/sgpr", bpc,
               fmas / ms / 1e9, 2 * fmas / ms / 1e9, ms);
      }
  const long long n = 1ll << 27;  // 2 GiB per array of double2? (n double2 = 2 GiB)
  double2 *a, *b, *c; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
  hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
  for (int blocks : {2048, 8192, 32768})
    {
      for (int rep = 0; rep < 3; ++rep)
        { hipEventRecord(e0); copy_kernel<<<blocks, 256>>>(c, a, n); hipEventRecord(e1); hipEventSynchronize(e1); }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("copy  blocks=%d: %.2f TB/s\n", blocks, 2.0 * n * 16 / ms / 1e9);
      for (int rep = 0; rep < 3; ++rep)
        { hipEventRecord(e0); triad_kernel<<<blocks, 256>>>(c, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); }
      hipEventElapsedTime(&ms, e0, e1);
      printf("triad blocks=%d: %.2f TB/s\n", blocks, 3.0 * n * 16 / ms / 1e9);
    }
  // the same copy with 4 loads in flight per lane and streaming accesses, at several working-set sizes (a set that
  // fits the 256 MiB Infinity Cache is served on-die: not an HBM rate)
  for (long long m : {1ll << 23, 1ll << 25, 1ll << 27})
    for (int blocks : {4096, 16384})
      {
        for (int rep = 0; rep < 5; ++rep)
          { hipEventRecord(e0); copy4_kernel<<<blocks, 256>>>((v2d *)c, (const v2d *)a, m); hipEventRecord(e1); hipEventSynchronize(e1); }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy4 (x4 unrolled, nontemporal) %5lld MiB per array, blocks=%d: %.2f TB/s\n", m * 16 >> 20, blocks, 2.0 * m * 16 / ms / 1e9);
      }
  return 0;
}
