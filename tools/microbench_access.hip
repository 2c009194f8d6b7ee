// Micro-benchmark: what does the access pattern of the tiled kernels' store phase cost?
// dst = b - src over a 321 x 321 x 321 DoF box (pitch 328) with
//   pattern 0: the tiled kernels' owner pattern (workgroup = 8 waves = 12 x 8 cells, a lane owns
//              4 consecutive DoFs of one row of 4 planes: 16-byte accesses at 32-byte stride, z-marching)
//   pattern 1: the same bytes per lane and per workgroup, but every wave instruction covers 1 KB
//              of consecutive addresses
// and 1, 2 or 4 workgroups per CU (limited through the dynamic LDS size).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));

constexpr int NP = 321, PITCH = 328;
constexpr long long PLANE = (long long)PITCH * NP;

template <int PATTERN, int DEPTH>
__global__ void __launch_bounds__(512) sweep(double *__restrict__ dst, const double *__restrict__ b, const double *__restrict__ src,
                                            int ntx, int nty, int zlen, int zmod)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tx = blockIdx.x % ntx, ty = (blockIdx.x / ntx) % nty, zc = blockIdx.x / (ntx * nty);
  if (PATTERN == 0)
    {
      const int cl = lane / 5, sub = lane % 5;
      const int cx = tx * 11 + cl - 1, cy = ty * 7 + w - 1;
      const bool own = lane < 60 && cl >= 1 && w >= 1 && sub < 4 && cx < 80 && cy < 80;
      const long long rowoff = own ? (long long)(4 * cy + sub) * PITCH + 4 * cx : 0;
      for (int cz = zc * zlen; cz < min((zc + 1) * zlen, 80); ++cz)
        {
          dbl2 x[DEPTH][2], y[DEPTH][2];
#pragma unroll
          for (int k = 0; k < DEPTH; ++k)
            {
              const long long off = (4ll * (cz % zmod) + k) * PLANE + rowoff;
              x[k][0] = *(const dbl2 *)(b + off);
              x[k][1] = *(const dbl2 *)(b + off + 2);
              y[k][0] = *(const dbl2 *)(src + off);
              y[k][1] = *(const dbl2 *)(src + off + 2);
            }
          if (own)
            {
#pragma unroll
              for (int k = 0; k < DEPTH; ++k)
                {
                  const long long off = (4ll * (cz % zmod) + k) * PLANE + rowoff;
                  *(dbl2 *)(dst + off)     = x[k][0] - y[k][0];
                  *(dbl2 *)(dst + off + 2) = x[k][1] - y[k][1];
                }
            }
        }
    }
  else
    {
      // every workgroup streams a contiguous chunk: 8 waves x 64 lanes x 16 B per instruction pair
      const long long per_wg_layer = 512ll * 2 * DEPTH * 2; // doubles per workgroup per "layer"
      const long long base         = ((long long)blockIdx.x * zlen) * per_wg_layer;
      for (int cz = 0; cz < zlen; ++cz)
        {
          dbl2 x[DEPTH][2], y[DEPTH][2];
          const long long o = base + (cz % zmod) * per_wg_layer + w * (per_wg_layer / 8);
#pragma unroll
          for (int k = 0; k < DEPTH; ++k)
            {
              const long long off = o + k * 256 + lane * 2;
              x[k][0] = *(const dbl2 *)(b + off);
              x[k][1] = *(const dbl2 *)(b + off + 128);
              y[k][0] = *(const dbl2 *)(src + off);
              y[k][1] = *(const dbl2 *)(src + off + 128);
            }
#pragma unroll
          for (int k = 0; k < DEPTH; ++k)
            {
              const long long off = o + k * 256 + lane * 2;
              *(dbl2 *)(dst + off)       = x[k][0] - y[k][0];
              *(dbl2 *)(dst + off + 128) = x[k][1] - y[k][1];
            }
        }
    }
}

int main()
{
  const long long n = 96ll * 80 * 8192 + 1024; // covers both patterns
  double *dst, *b, *src;
  hipMalloc(&dst, n * 8);
  hipMalloc(&b, n * 8);
  hipMalloc(&src, n * 8);
  hipMemset(b, 0, n * 8);
  hipMemset(src, 0, n * 8);
  hipMemset(dst, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int ntx = 8, nty = 12;
  for (int zmod : {1000, 2, 1})
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int wgs_per_cu : {1, 2})
      for (int nzc : {5})
        {
          const int    zlen = 80 / nzc;
          const size_t lds  = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 60 * 1024 : 30 * 1024);
          auto         k    = pattern == 0 ? sweep<0, 4> : sweep<1, 4>;
          hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          float best = 1e9;
          for (int rep = 0; rep < 5; ++rep)
            {
              hipEventRecord(e0);
              hipLaunchKernelGGL(k, dim3(ntx * nty * nzc), dim3(512), lds, 0, dst, b, src, ntx, nty, zlen, zmod);
              hipEventRecord(e1);
              hipEventSynchronize(e1);
              float ms;
              hipEventElapsedTime(&ms, e0, e1);
              best = ms < best ? ms : best;
            }
          // bytes: pattern 0 moves 24 B per owned DoF (80^3 cells x 64); pattern 1 per lane
          const double bytes = pattern == 0 ? 24.0 * 80 * 80 * 80 * 64 : 24.0 * (double)ntx * nty * nzc * zlen * 512 * 16;
          printf("zmod=%d pattern=%s wg/CU<=%d chunks=%d: %.3f ms  %.2f TB/s\n", zmod, pattern == 0 ? "tile " : "linear", wgs_per_cu, nzc, best,
                 bytes / best / 1e9);
        }
  return 0;
}
