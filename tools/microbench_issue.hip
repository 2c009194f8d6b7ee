// Issue-rate micro-benchmark: one wave per SIMD (40 KB of LDS per 64-thread block), chains of fp64 FMAs with ILP
// independent accumulators, optionally interleaved with accumulation-register moves.  Prints cycles per instruction
// (s_memtime) and the chip-wide FMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, int MOVES, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) fma_kernel(double *out, const double *in, int iters, unsigned long long *cyc)
{
  extern __shared__ char lds[];
  double acc[ILP];
  const double a = in[threadIdx.x], b = in[threadIdx.x + 64];
#pragma unroll
  for (int i = 0; i < ILP; ++i)
    acc[i] = in[threadIdx.x + 128 + i];
  int park[MOVES > 0 ? MOVES : 1];
#pragma unroll
  for (int m = 0; m < MOVES; ++m)
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[m]) : "v"(m));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
        {
#pragma unroll
          for (int i = 0; i < ILP; ++i)
            acc[i] = fma(acc[i], a, b);
#pragma unroll
          for (int m = 0; m < MOVES; ++m)
            {
              int t;
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(park[m]));
              asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[m]) : "v"(t));
            }
        }
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0.;
#pragma unroll
  for (int i = 0; i < ILP; ++i)
    s += acc[i];
  int ps = 0;
#pragma unroll
  for (int m = 0; m < MOVES; ++m)
    {
      int t;
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(park[m]));
      ps += t;
    }
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s + ps;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    cyc[0] = t1 - t0;
  if (lds[threadIdx.x] == 77)
    out[0] = 1.;
}

// FMA chains whose multiplier / addend are wave-uniform values in scalar registers (the sweeps' coefficient operands),
// KIND 0: acc = fma(acc, s, b)   1: acc = fma(s, x, acc) with 8 different scalars   2: acc = acc * s   3: acc = acc + b (vector)
template <int ILP, int KIND>
__global__ void __launch_bounds__(64) sfma_kernel(double *out, const double *in, const double *coef, int iters, unsigned long long *cyc)
{
  extern __shared__ char lds[];
  const __attribute__((address_space(4))) double *cs = (const __attribute__((address_space(4))) double *)coef;
  double sc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    sc[i] = cs[i];
  double acc[ILP];
  const double b = in[threadIdx.x + 64], x = in[threadIdx.x];
#pragma unroll
  for (int i = 0; i < ILP; ++i)
    acc[i] = in[threadIdx.x + 128 + i];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < ILP; ++i)
          {
            if (KIND == 0)
              acc[i] = fma(acc[i], sc[0], b);
            else if (KIND == 1)
              acc[i] = fma(sc[(i + rep) & 7], x, acc[i]);
            else if (KIND == 2)
              acc[i] = acc[i] * sc[(i + rep) & 7];
            else
              acc[i] = acc[i] + b;
          }
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0.;
#pragma unroll
  for (int i = 0; i < ILP; ++i)
    s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    cyc[0] = t1 - t0;
  if (lds[threadIdx.x] == 77)
    out[0] = 1.;
}
template <int ILP, int KIND>
void srun(const char *name, double *out, double *in, unsigned long long *cyc)
{
  const int iters = 2000, blocks = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  sfma_kernel<ILP, KIND><<<blocks, 64, 40 * 1024>>>(out, in, in + 2048, iters, cyc);
  hipEventRecord(e0);
  sfma_kernel<ILP, KIND><<<blocks, 64, 40 * 1024>>>(out, in, in + 2048, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s ILP %2d: %6.2f cycles/instr, %7.3f ms\n", name, ILP, (double)c / ((double)iters * 8 * ILP), ms);
}

template <int ILP, int MOVES, int WAVES>
void run(const char *name, double *out, double *in, unsigned long long *cyc, int lds_bytes, int blocks)
{
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)fma_kernel<ILP, MOVES, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  fma_kernel<ILP, MOVES, WAVES><<<blocks, 64 * WAVES, lds_bytes>>>(out, in, iters, cyc);
  hipEventRecord(e0);
  fma_kernel<ILP, MOVES, WAVES><<<blocks, 64 * WAVES, lds_bytes>>>(out, in, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double ninst = (double)iters * 8 * (ILP + 2 * MOVES);
  printf("%-34s ILP %2d moves %2d waves/SIMD %d: %6.2f cycles/instr (s_memtime), %7.3f ms, %6.2f TFMA/s chip\n", name, ILP, MOVES, WAVES / 4 ? WAVES / 4 : 1,
         (double)c / ninst, ms, (double)blocks * 64 * WAVES * iters * 8 * ILP / ms / 1e9);
}

int main()
{
  double *out, *in;
  unsigned long long *cyc;
  hipMalloc(&out, 1024 * 512 * 8);
  hipMalloc(&in, 4096 * 8);
  hipMalloc(&cyc, 8);
  hipMemset(in, 0, 4096 * 8);
  // one wave per SIMD: 64-thread blocks with 40 KB of LDS (4 per CU)
  run<1, 0, 1>("1 wave/SIMD dependent chain", out, in, cyc, 40 * 1024, 1024);
  run<2, 0, 1>("1 wave/SIMD", out, in, cyc, 40 * 1024, 1024);
  run<4, 0, 1>("1 wave/SIMD", out, in, cyc, 40 * 1024, 1024);
  run<8, 0, 1>("1 wave/SIMD", out, in, cyc, 40 * 1024, 1024);
  run<16, 0, 1>("1 wave/SIMD", out, in, cyc, 40 * 1024, 1024);
  run<8, 4, 1>("1 wave/SIMD + acc moves", out, in, cyc, 40 * 1024, 1024);
  run<8, 8, 1>("1 wave/SIMD + acc moves", out, in, cyc, 40 * 1024, 1024);
  srun<8, 0>("fma(acc, SGPR, vgpr)", out, in, cyc);
  srun<8, 1>("fma(SGPR_i, vgpr, acc)  [8 scalars]", out, in, cyc);
  srun<16, 1>("fma(SGPR_i, vgpr, acc)  [8 scalars]", out, in, cyc);
  srun<8, 2>("acc * SGPR_i", out, in, cyc);
  srun<8, 3>("acc + vgpr", out, in, cyc);
  srun<16, 3>("acc + vgpr", out, in, cyc);
  // two waves per SIMD: 512-thread blocks (8 waves), one per CU
  run<1, 0, 8>("2 waves/SIMD dependent chain", out, in, cyc, 100 * 1024, 256);
  run<4, 0, 8>("2 waves/SIMD", out, in, cyc, 100 * 1024, 256);
  run<8, 0, 8>("2 waves/SIMD", out, in, cyc, 100 * 1024, 256);
  return 0;
}
