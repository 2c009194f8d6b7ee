// LDS micro-benchmark for the wave-pair sweeps (one wave per SIMD, two workgroups of two waves per compute unit, as in
// csrc/kernels_pair.hip): what a z column's scratchpad traffic costs next to its arithmetic.  Per loop iteration a wave
// issues NR reads and NW writes (KIND 0: ds_read2st64_b64 / ds_write2st64_b64 of lane-strided doubles, the scratchpad's
// layout; 1: ds_read_b128 / ds_write_b128 of 16 contiguous bytes per lane; 2: ds_read_b64 / ds_write_b64) and NF
// independent fp64 FMAs (8 chains); a read is waited for one iteration after it was issued.  Prints cycles per iteration
// (s_memtime of wave 0 of workgroup 0) with every compute unit loaded (512 workgroups) and with one workgroup alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int NR, int NW, int NF, int KIND>
__global__ void __launch_bounds__(128) lds_kernel(double *out, const double *in, int iters, unsigned long long *cyc)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 9000; i += 128)
    lds[i] = in[i & 255];
  __syncthreads();
  // byte address of this lane's slot 0 (lane-strided: slot s at (s * 64 + lane) doubles; contiguous: lane * 16 B + s KB)
  const unsigned base = (KIND == 0 || KIND == 2) ? (wave * 3200 + lane) * 8u : wave * 25600u + lane * 16u;
  double         acc[8];
  const double   a = in[threadIdx.x], b = in[threadIdx.x + 64];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    acc[i] = in[threadIdx.x + 128 + i];
  v4u r[NR > 0 ? NR : 1];
  v2u q[NR > 0 ? NR : 1];
#pragma unroll
  for (int k = 0; k < (NR > 0 ? NR : 1); ++k)
    {
      r[k] = v4u{0u, 0u, 0u, 0u};
      q[k] = v2u{0u, 0u};
    }
  v4u w = {1u, 2u, 3u, 4u};
  // (the operand loads have completed before the clock is read: the compiler may reuse their base registers for it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 4" ::"v"(a), "v"(b), "v"(acc[0]), "v"(acc[7]));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int k = 0; k < NR; ++k)
        {
          if (KIND == 0)
            asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r[k]) : "v"(base), "n"(2 * k), "n"(2 * k + 25));
          else if (KIND == 1)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(base), "n"(1024 * k));
          else
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[k]) : "v"(base), "n"(512 * k));
        }
#pragma unroll
      for (int rep = 0; rep < NF / 8; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i] = fma(acc[i], a, b);
#pragma unroll
      for (int k = 0; k < NW; ++k)
        {
          if (KIND == 0)
            asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base), "v"(v2u{w.x, w.y}), "v"(v2u{w.z, w.w}), "n"(2 * k + 1), "n"(2 * k + 26));
          else if (KIND == 1)
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base), "v"(w), "n"(1024 * k + 12288));
          else
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base), "v"(v2u{w.x, w.y}), "n"(512 * k + 12288));
        }
      // everything of the iteration before this one has completed
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR + NW < 16 ? NR + NW : 15));
    }
  asm volatile("s_waitcnt lgkmcnt(0)");
  // (the reads' destination registers stay allocated until the wait above: the compiler takes an asm output for
  //  available at once and would otherwise reuse the unused parts of a destination for, e.g., the address of `out`
  //  while the read is still in flight - the first version of this file faulted that way)
#pragma unroll
  for (int k = 0; k < NR; ++k)
    asm volatile("" : "+v"(r[k]), "+v"(q[k]));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double                   s = 0.;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += acc[i];
#pragma unroll
  for (int k = 0; k < NR; ++k)
    s += (double)r[k].x + (double)q[k].x;
  out[blockIdx.x * 128 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    cyc[0] = t1 - t0;
}

template <int NR, int NW, int NF, int KIND>
void
run(const char *what, double *out, const double *in, unsigned long long *cyc)
{
  const int    iters = 2000;
  const size_t lds   = 77 * 1024; // two workgroups per compute unit
  hipFuncSetAttribute((const void *)lds_kernel<NR, NW, NF, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  double res[2];
  int    k = 0;
  for (const int blocks : {512, 1})
    {
      lds_kernel<NR, NW, NF, KIND><<<blocks, 128, lds>>>(out, in, iters, cyc);
      lds_kernel<NR, NW, NF, KIND><<<blocks, 128, lds>>>(out, in, iters, cyc);
      if (hipDeviceSynchronize() != hipSuccess)
        {
          printf("%s: kernel failed\n", what);
          fflush(stdout);
          exit(1);
        }
      unsigned long long c = 0;
      hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      res[k++] = (double)c / iters;
    }
  const double bytes = (KIND == 0 ? 1024. : KIND == 1 ? 1024. : 512.) * (NR + NW);
  printf("%-44s reads %d writes %d fma %2d: %7.1f cycles/iteration loaded (%5.1f B/clk per CU from 4 waves), %7.1f alone\n", what, NR, NW, NF, res[0],
         4. * bytes / res[0], res[1]);
  fflush(stdout);
}

int
main()
{
  double *            out, *in;
  unsigned long long *cyc;
  hipMalloc(&out, 512 * 128 * sizeof(double));
  hipMalloc(&in, 4096 * sizeof(double));
  hipMalloc(&cyc, 8);
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i)
    h[i] = 1. / (1 + i % 97);
  hipMemcpy(in, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
  run<0, 0, 40, 0>("arithmetic only", out, in, cyc);
  run<2, 2, 0, 0>("read2st64_b64 / write2st64_b64", out, in, cyc);
  run<2, 2, 40, 0>("read2st64_b64 / write2st64_b64", out, in, cyc);
  run<4, 0, 0, 0>("read2st64_b64", out, in, cyc);
  run<0, 4, 0, 0>("write2st64_b64", out, in, cyc);
  run<4, 0, 40, 0>("read2st64_b64", out, in, cyc);
  run<0, 4, 40, 0>("write2st64_b64", out, in, cyc);
  run<5, 2, 40, 0>("read2st64_b64 / write2st64_b64", out, in, cyc);
  run<8, 0, 0, 0>("read2st64_b64", out, in, cyc);
  run<2, 2, 0, 1>("read_b128 / write_b128", out, in, cyc);
  run<2, 2, 40, 1>("read_b128 / write_b128", out, in, cyc);
  run<4, 0, 0, 1>("read_b128", out, in, cyc);
  run<0, 4, 0, 1>("write_b128", out, in, cyc);
  run<4, 4, 0, 2>("read_b64 / write_b64", out, in, cyc);
  run<4, 4, 40, 2>("read_b64 / write_b64", out, in, cyc);
  run<8, 0, 0, 2>("read_b64", out, in, cyc);
  run<0, 8, 0, 2>("write_b64", out, in, cyc);
  return 0;
}
