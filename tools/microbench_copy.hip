// Micro-benchmark: what copy rate do these boxes sustain?  (bytes read + bytes written) / time, arrays far beyond the
// 256 MiB Infinity Cache.  Variants: U 16-byte loads in flight per lane before the first store, plain / non-temporal
// accesses, B workgroups of 256 threads per compute unit, contiguous 1 KB per wave instruction; every variant checks its
// own bounds (the copy4 line of tools/microbench.hip with 16384 blocks on 128 MiB did not run a single iteration).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_kernel(v4f *__restrict__ dst, const v4f *__restrict__ src, const long long n)
{
  // a block takes tiles of U * 256 vectors; the U loads of a lane are 256 vectors (4 KB) apart
  const long long tile = 256ll * U;
  for (long long t = blockIdx.x; t * tile < n; t += gridDim.x)
    {
      const long long i = t * tile + threadIdx.x;
      v4f             v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i + 256ll * u < n)
          v[u] = NT ? __builtin_nontemporal_load(src + i + 256ll * u) : src[i + 256ll * u];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i + 256ll * u < n)
          {
            if (NT)
              __builtin_nontemporal_store(v[u], dst + i + 256ll * u);
            else
              dst[i + 256ll * u] = v[u];
          }
    }
}

template <int U, bool NT>
static double run(v4f *dst, const v4f *src, long long n, int blocks, hipEvent_t e0, hipEvent_t e1)
{
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep)
    {
      hipEventRecord(e0);
      hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(blocks), dim3(256), 0, 0, dst, src, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best)
        best = ms;
    }
  return 2.0 * n * 16 / best / 1e9;
}

int main()
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (long long mib : {1024ll, 4096ll})
    {
      const long long n = mib * (1 << 20) / 16;
      v4f *a, *b;
      if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess)
        return 1;
      hipMemset(a, 1, n * 16);
      hipMemset(b, 0, n * 16);
      for (int bpc : {2, 4, 8, 16, 32})
        {
          const int blocks = 256 * bpc;
          printf("%4lld MiB/array, %2d blocks/CU:  U=1 %.2f  U=2 %.2f  U=4 %.2f  U=8 %.2f | nt: U=1 %.2f  U=2 %.2f  U=4 %.2f  U=8 %.2f TB/s\n", mib, bpc,
                 run<1, false>(b, a, n, blocks, e0, e1), run<2, false>(b, a, n, blocks, e0, e1), run<4, false>(b, a, n, blocks, e0, e1),
                 run<8, false>(b, a, n, blocks, e0, e1), run<1, true>(b, a, n, blocks, e0, e1), run<2, true>(b, a, n, blocks, e0, e1),
                 run<4, true>(b, a, n, blocks, e0, e1), run<8, true>(b, a, n, blocks, e0, e1));
        }
      // one tile per block (no loop): as many blocks as tiles
      printf("%4lld MiB/array, one tile per block: U=4 %.2f  nt U=4 %.2f  nt U=8 %.2f TB/s\n", mib,
             run<4, false>(b, a, n, (int)(n / 1024), e0, e1), run<4, true>(b, a, n, (int)(n / 1024), e0, e1), run<8, true>(b, a, n, (int)(n / 2048), e0, e1));
      hipFree(a);
      hipFree(b);
    }
  return 0;
}
