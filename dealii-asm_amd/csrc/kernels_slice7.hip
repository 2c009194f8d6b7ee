// Slice-per-lane patch solve for 7^3 patches (degree 4: n_overlap = 2 and vertex-star) on the region of a
// tensor-product mesh whose cells share one parity-sorted FDM instance per direction (every cell that is not
// next to a boundary: 93 % of an 80^3 Dirichlet box, all of a periodic one).
//
// It replaces the patch solve of fdm_generic_kernel<7> there (ASPoissonPreconditioner::vmult_internal,
// include/matrix_free.h:1066-1162, with TensorProductMatrixSymmetricSum::apply_inverse): same inputs, same
// output (the weighted patch result in tmp[cell][k][j][i], summed per DoF by fdm_gather_kernel), but in the
// register layout of the tiled 5^3 kernels instead of one thread per line with the tile in LDS:
//   * 7 lanes per patch, 9 patches (consecutive in x) per wave; a lane holds one 7 x 7 slice of the patch tile in
//     98 registers: layout A lane = z plane k, v[j*7 + i]; layout B lane = y index, v[k*7 + i];
//   * the x and y sweeps (layout A) and the z sweep (layout B) are register FMA chains in even-odd form
//     (16 + 9 products and 6 sums per 7-point line instead of 49 products) with wave-uniform scalar operands;
//   * two 7-lane transposes through a wave-private LDS buffer; no workgroup barrier anywhere: waves are independent.
// The generic kernel needs 12 LDS passes over the tile and 6 barriers per patch; this one 2 transposes.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace asmamd
{
  namespace
  {
    constexpr int M   = 7;
    constexpr int M2  = 49;
    constexpr int M3  = 343;
    constexpr int H   = 3;  // M / 2
    constexpr int NEV = 4;  // even eigenvectors
    constexpr int E   = M2 + M; // eigen data per instance
    constexpr int PPW = S7_TILE_X; // patches per wave
    constexpr int WPB = 4;  // waves per workgroup
    constexpr int AW  = S7_TILE_Y; // waves per workgroup of the accumulate form
    constexpr int TB7 = 64 * M; // doubles of a wave's transpose buffer

    struct UniformMatrix7
    {
      const __attribute__((address_space(4))) double *p;
      __device__ __forceinline__ explicit UniformMatrix7(const double *q)
        : p((const __attribute__((address_space(4))) double *)q)
      {}
      __device__ __forceinline__ double operator[](const int e) const { return p[e]; }
    };

    // out[a] = sum_i Q[i][a] in[i]; columns 0..3 even, 4..6 odd vectors
    __device__ __forceinline__ void
    line7_fwd(double &u0, double &u1, double &u2, double &u3, double &u4, double &u5, double &u6, const UniformMatrix7 &Q)
    {
      const double e0 = u0 + u6, e1 = u1 + u5, e2 = u2 + u4, e3 = u3;
      const double o0 = u0 - u6, o1 = u1 - u5, o2 = u2 - u4;
      double       t[M];
#pragma unroll
      for (int a = 0; a < NEV; ++a)
        t[a] = fma(Q[3 * M + a], e3, fma(Q[2 * M + a], e2, fma(Q[1 * M + a], e1, Q[0 * M + a] * e0)));
#pragma unroll
      for (int a = NEV; a < M; ++a)
        t[a] = fma(Q[2 * M + a], o2, fma(Q[1 * M + a], o1, Q[0 * M + a] * o0));
      u0 = t[0];
      u1 = t[1];
      u2 = t[2];
      u3 = t[3];
      u4 = t[4];
      u5 = t[5];
      u6 = t[6];
    }
    // out[i] = sum_a Q[i][a] in[a]
    __device__ __forceinline__ void
    line7_bwd(double &u0, double &u1, double &u2, double &u3, double &u4, double &u5, double &u6, const UniformMatrix7 &Q)
    {
      double Ev[NEV], Od[H];
#pragma unroll
      for (int i = 0; i < NEV; ++i)
        Ev[i] = fma(Q[i * M + 3], u3, fma(Q[i * M + 2], u2, fma(Q[i * M + 1], u1, Q[i * M + 0] * u0)));
#pragma unroll
      for (int i = 0; i < H; ++i)
        Od[i] = fma(Q[i * M + 6], u6, fma(Q[i * M + 5], u5, Q[i * M + 4] * u4));
      u0 = Ev[0] + Od[0];
      u6 = Ev[0] - Od[0];
      u1 = Ev[1] + Od[1];
      u5 = Ev[1] - Od[1];
      u2 = Ev[2] + Od[2];
      u4 = Ev[2] - Od[2];
      u3 = Ev[3];
    }
    template <bool FWD>
    __device__ __forceinline__ void
    sweep_fast7(double (&v)[M2], const UniformMatrix7 &Q)
    {
#pragma unroll
      for (int s = 0; s < M; ++s)
        {
          if (FWD)
            line7_fwd(v[s * M + 0], v[s * M + 1], v[s * M + 2], v[s * M + 3], v[s * M + 4], v[s * M + 5], v[s * M + 6], Q);
          else
            line7_bwd(v[s * M + 0], v[s * M + 1], v[s * M + 2], v[s * M + 3], v[s * M + 4], v[s * M + 5], v[s * M + 6], Q);
        }
    }
    template <bool FWD>
    __device__ __forceinline__ void
    sweep_slow7(double (&v)[M2], const UniformMatrix7 &Q)
    {
#pragma unroll
      for (int f = 0; f < M; ++f)
        {
          if (FWD)
            line7_fwd(v[0 * M + f], v[1 * M + f], v[2 * M + f], v[3 * M + f], v[4 * M + f], v[5 * M + f], v[6 * M + f], Q);
          else
            line7_bwd(v[0 * M + f], v[1 * M + f], v[2 * M + f], v[3 * M + f], v[4 * M + f], v[5 * M + f], v[6 * M + f], Q);
        }
    }

    // 7-lane transpose: lane `sub` of a patch holds v[s*7 + f]; afterwards v'[s*7 + f] = (v[sub*7 + f] of the patch's lane s).
    // The LDS executes a wave's accesses in order: no wait between the rounds (see transpose5 in kernels_tiled.hip).
    __device__ __forceinline__ void
    transpose7(double (&v)[M2], double *__restrict__ Tw, const int lane, const int cl, const int sub)
    {
#pragma unroll
      for (int f = 0; f < M; ++f)
        {
#pragma unroll
          for (int s = 0; s < M; ++s)
            Tw[lane * M + s] = v[s * M + f];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < M; ++s)
            v[s * M + f] = Tw[(cl * M + s) * M + sub];
          asm volatile("" ::: "memory");
        }
    }

    struct Slice7Args
    {
      int           lo[3], n[3];  // region of cells [lo, lo + n)
      int           npx;          // waves along x
      const double *qx, *qy, *qz; // the region's eigenvector matrices (E doubles each)
      const double *inv;          // [c][b][a] reciprocal eigenvalue sums of the region's instance triple
      // accumulate form: columns of PPW x WPB cells, z chunks of lz layers; flag[d][storage index] = 0: only one column /
      // chunk of the box touches the DoF plane in direction d
      int                  ntx, nty, nzc, lz;
      const unsigned char *flag[3];
    };

    // gather + solve of one wave's patches on the uniform box; result in layout A: lane = x index `sub`, v[k*7 + j]
    template <bool HAS_PRE, bool HAS_POST>
    __device__ __forceinline__ void
    solve_uniform(double (&v)[M2], const DevGrid &g, const DevFdm &f, const Slice7Args &sa, const double *__restrict__ src, const int cx,
                  const int cy, const int cz, const int lane, const int cl, const int sub, double *__restrict__ Tw)
    {
      const int o0 = f.origin_mul * cx + f.origin_add, o1 = f.origin_mul * cy + f.origin_add, o2 = f.origin_mul * cz + f.origin_add;
      // every DoF of a region patch exists (the region's cells have both neighbours in every direction)
      // Orientation: the 7 lanes of a patch are its x indices, the registers v[k*7 + j] its (z, y) plane - the lanes of a
      // patch then read 7 consecutive DoFs of src and write 7 consecutive entries of tmp with every instruction (with the
      // lanes along z, as in round 2's first version, every lane touched its own 64-byte sector).
      const long long xoff = g.store_index_near(0, o0 + sub);
      long long       yoff[M];
#pragma unroll
      for (int jj = 0; jj < M; ++jj)
        yoff[jj] = g.store_index_near(1, o1 + jj) * g.stride[1];

      // ---- gather: layout A, lane = x index `sub`, v[k*7 + j]
#pragma unroll
      for (int k = 0; k < M; ++k)
        {
          const double *plane = src + xoff + g.store_index_near(2, o2 + k) * g.stride[2];
#pragma unroll
          for (int jj = 0; jj < M; ++jj)
            v[k * M + jj] = plane[yoff[jj]];
        }
      if (HAS_PRE)
        {
          // the rows of weight factors are the same for every cell of the region: their y-z products are scalar
          // operands, the x factor belongs to the lane
          const UniformMatrix7 wyz(f.s7_wyz[0]);
          const double         wx = f.w_pre[0][(sa.lo[0] + 1) * M + sub];
#pragma unroll
          for (int e = 0; e < M2; ++e)
            v[e] = (v[e] * wyz[e]) * wx;
        }
      const UniformMatrix7 Qx(sa.qx), Qy(sa.qy), Qz(sa.qz);
      sweep_fast7<true>(v, Qy);  // j -> b
      sweep_slow7<true>(v, Qz);  // k -> c
      transpose7(v, Tw, lane, cl, sub); // layout B: lane = c, v[i*7 + b]
      sweep_slow7<true>(v, Qx);  // i -> a
      {
        const double *inv = sa.inv + sub * M; // [a][c = sub][b]
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
          for (int b = 0; b < M; ++b)
            v[a * M + b] *= inv[a * M2 + b];
      }
      sweep_slow7<false>(v, Qx);
      transpose7(v, Tw, lane, cl, sub); // layout A: lane = i, v[c*7 + b]
      sweep_slow7<false>(v, Qz);
      sweep_fast7<false>(v, Qy);
      if (HAS_POST)
        {
          const UniformMatrix7 wyz(f.s7_wyz[1]);
          const double         wx = f.w_post[0][(sa.lo[0] + 1) * M + sub];
#pragma unroll
          for (int e = 0; e < M2; ++e)
            v[e] = (v[e] * wyz[e]) * wx;
        }
    }

    template <bool HAS_PRE, bool HAS_POST>
    __global__ void __launch_bounds__(WPB * 64, 3)
    fdm_slice7_kernel(const DevGrid g, const DevFdm f, const Slice7Args sa, const double *__restrict__ src, double *__restrict__ tmp)
    {
      __shared__ double S_T[WPB * TB7];
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      const long long gw = xcd_block(blockIdx.x, gridDim.x) * WPB + w; // wave index: (px, cy, cz)
      const long long n_waves = (long long)sa.npx * sa.n[1] * sa.n[2];
      if (gw >= n_waves)
        return; // no barrier in this kernel
      const int px = (int)(gw % sa.npx), cy = sa.lo[1] + (int)((gw / sa.npx) % sa.n[1]), cz = sa.lo[2] + (int)(gw / ((long long)sa.npx * sa.n[1]));
      int       cl = lane / M;
      const int sub = lane - cl * M;
      int       cx  = sa.lo[0] + px * PPW + cl;
      // lanes without a patch (lane 63, patches beyond the region) run along on the last valid patch and store nothing
      const bool active = cl < PPW && cx < sa.lo[0] + sa.n[0];
      if (!active)
        cx = sa.lo[0] + sa.n[0] - 1;
      double v[M2];
      solve_uniform<HAS_PRE, HAS_POST>(v, g, f, sa, src, cx, cy, cz, lane, cl, sub, S_T + w * TB7);
      if (active)
        {
          const long long cell = cx + (long long)g.nc[0] * (cy + (long long)g.nc[1] * cz);
          double *        out  = tmp + cell * M3 + sub; // tmp[cell][k][j][i]
#pragma unroll
          for (int e = 0; e < M2; ++e)
            out[e * M] = v[e];
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // The same register layout for meshes WITHOUT tensor-product structure (deformed cells: every cell has its own three
    // 1-D eigen-decompositions, include/matrix_free.h:1023-1162 with one TensorProductMatrixSymmetricSum per cell): the
    // 7 x 7 matrix of the direction being swept is held in registers by the 7 lanes of the patch (one copy per lane, the
    // loads of a patch's lanes coalesce), full 7-point lines (no parity structure), reciprocal eigenvalue sums computed
    // on the fly.  Five matrix loads per patch (the x matrix serves its forward and backward sweep).
    // ---------------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void
    load_matrix7(double (&q)[M2], const double *__restrict__ p)
    {
#pragma unroll
      for (int e = 0; e < M2; ++e)
        q[e] = p[e];
    }
    // line: out[a] = sum_i Q[i][a] in[i] (FWD) / out[i] = sum_a Q[i][a] in[a]
    template <bool FWD>
    __device__ __forceinline__ void
    line7_reg(double &u0, double &u1, double &u2, double &u3, double &u4, double &u5, double &u6, const double (&Q)[M2])
    {
      const double in[M] = {u0, u1, u2, u3, u4, u5, u6};
      double       out[M];
#pragma unroll
      for (int a = 0; a < M; ++a)
        {
          double t = (FWD ? Q[a] : Q[a * M]) * in[0];
#pragma unroll
          for (int i = 1; i < M; ++i)
            t = fma(FWD ? Q[i * M + a] : Q[a * M + i], in[i], t);
          out[a] = t;
        }
      u0 = out[0];
      u1 = out[1];
      u2 = out[2];
      u3 = out[3];
      u4 = out[4];
      u5 = out[5];
      u6 = out[6];
    }
    template <bool FWD>
    __device__ __forceinline__ void
    sweep_fast7_reg(double (&v)[M2], const double (&Q)[M2])
    {
#pragma unroll
      for (int s = 0; s < M; ++s)
        line7_reg<FWD>(v[s * M + 0], v[s * M + 1], v[s * M + 2], v[s * M + 3], v[s * M + 4], v[s * M + 5], v[s * M + 6], Q);
    }
    template <bool FWD>
    __device__ __forceinline__ void
    sweep_slow7_reg(double (&v)[M2], const double (&Q)[M2])
    {
#pragma unroll
      for (int f = 0; f < M; ++f)
        line7_reg<FWD>(v[0 * M + f], v[1 * M + f], v[2 * M + f], v[3 * M + f], v[4 * M + f], v[5 * M + f], v[6 * M + f], Q);
    }

    // gather + solve of one wave's patches with per-cell eigen-decompositions; result in layout A as solve_uniform
    template <bool HAS_PRE, bool HAS_POST>
    __device__ __forceinline__ void
    solve_percell(double (&v)[M2], const DevGrid &g, const DevFdm &f, const double *__restrict__ src, const int cx, const int cy, const int cz,
                  const int lane, const int cl, const int sub, double *__restrict__ Tw)
    {
      const int o0 = f.origin_mul * cx + f.origin_add, o1 = f.origin_mul * cy + f.origin_add, o2 = f.origin_mul * cz + f.origin_add;
      const long long ce = (cx + 1) + (long long)(g.nc[0] + 2) * ((cy + 1) + (long long)(g.nc[1] + 2) * (cz + 1));
      const double *  ex = f.eig + (long long)f.inst[0][ce] * E, *ey = f.eig + (long long)f.inst[1][ce] * E,
                   *ez = f.eig + (long long)f.inst[2][ce] * E;
      // lanes along x, registers v[k*7 + j] (as in solve_uniform)
      const long long xoff = g.store_index_near(0, o0 + sub);
#pragma unroll
      for (int k = 0; k < M; ++k)
        {
          const double *plane = src + xoff + g.store_index_near(2, o2 + k) * g.stride[2];
#pragma unroll
          for (int jj = 0; jj < M; ++jj)
            v[k * M + jj] = plane[g.store_index_near(1, o1 + jj) * g.stride[1]];
        }
      if (HAS_PRE)
        {
          const double wx = f.w_pre[0][(cx + 1) * M + sub];
#pragma unroll
          for (int k = 0; k < M; ++k)
            {
              const double wzx = f.w_pre[2][(cz + 1) * M + k] * wx;
#pragma unroll
              for (int jj = 0; jj < M; ++jj)
                v[k * M + jj] *= f.w_pre[1][(cy + 1) * M + jj] * wzx;
            }
        }
      double q[M2];
      load_matrix7(q, ey);
      sweep_fast7_reg<true>(v, q); // j -> b
      load_matrix7(q, ez);
      sweep_slow7_reg<true>(v, q); // k -> c
      transpose7(v, Tw, lane, cl, sub); // layout B: lane = c, v[i*7 + b]
      load_matrix7(q, ex);
      sweep_slow7_reg<true>(v, q); // i -> a
      {
        const double lz = ez[M2 + sub];
#pragma unroll
        for (int b = 0; b < M; ++b)
          {
            const double lyz = ey[M2 + b] + lz;
#pragma unroll
            for (int a = 0; a < M; ++a)
              {
                // reciprocal with two Newton steps (full double precision)
                const double sm = ex[M2 + a] + lyz;
                double       r  = __builtin_amdgcn_rcp(sm);
                r               = fma(fma(-sm, r, 1.0), r, r);
                r               = fma(fma(-sm, r, 1.0), r, r);
                v[a * M + b] *= r;
              }
          }
      }
      sweep_slow7_reg<false>(v, q);     // (the x matrix is still in q)
      transpose7(v, Tw, lane, cl, sub); // layout A: lane = i, v[c*7 + b]
      load_matrix7(q, ez);
      sweep_slow7_reg<false>(v, q);
      load_matrix7(q, ey);
      sweep_fast7_reg<false>(v, q);
      if (HAS_POST)
        {
          const double wx = f.w_post[0][(cx + 1) * M + sub];
#pragma unroll
          for (int k = 0; k < M; ++k)
            {
              const double wzx = f.w_post[2][(cz + 1) * M + k] * wx;
#pragma unroll
              for (int jj = 0; jj < M; ++jj)
                v[k * M + jj] *= f.w_post[1][(cy + 1) * M + jj] * wzx;
            }
        }
    }

    template <bool HAS_PRE, bool HAS_POST>
    __global__ void __launch_bounds__(WPB * 64, 2)
    fdm_slice7_cell_kernel(const DevGrid g, const DevFdm f, const Slice7Args sa, const double *__restrict__ src, double *__restrict__ tmp)
    {
      __shared__ double S_T[WPB * TB7];
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      const long long gw = xcd_block(blockIdx.x, gridDim.x) * WPB + w; // wave index: (px, cy, cz)
      const long long n_waves = (long long)sa.npx * sa.n[1] * sa.n[2];
      if (gw >= n_waves)
        return; // no barrier in this kernel
      const int px = (int)(gw % sa.npx), cy = sa.lo[1] + (int)((gw / sa.npx) % sa.n[1]), cz = sa.lo[2] + (int)(gw / ((long long)sa.npx * sa.n[1]));
      int       cl = lane / M;
      const int sub = lane - cl * M;
      int       cx  = sa.lo[0] + px * PPW + cl;
      const bool active = cl < PPW && cx < sa.lo[0] + sa.n[0];
      if (!active)
        cx = sa.lo[0] + sa.n[0] - 1;
      double v[M2];
      solve_percell<HAS_PRE, HAS_POST>(v, g, f, src, cx, cy, cz, lane, cl, sub, S_T + w * TB7);
      if (active)
        {
          const long long cell = cx + (long long)g.nc[0] * (cy + (long long)g.nc[1] * cz);
          double *        out  = tmp + cell * M3 + sub; // tmp[cell][k][j][i]
#pragma unroll
          for (int e = 0; e < M2; ++e)
            out[e * M] = v[e];
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // Owner-computes form of the 7^3 sweep (cell_operation_overlap with its scatter-add, include/matrix_free.h:1066-1162,
    // read_write_operation.h:3-132) WITHOUT the tmp[cell][343] round trip (43 B/DoF written and read again):
    //   * a workgroup of WPB waves takes a column of PPW x WPB cells and marches along z over a chunk of sa.lz layers;
    //     wave w solves the PPW patches of cell row w of the layer, exactly as the kernels above;
    //   * the patch results are summed in an LDS accumulator (ds_add_f64) of 8 z planes x (4 WPB + 3) rows x (4 PPW + 3)
    //     DoFs, used as a ring along z: the patches of layer l cover planes 4 l ... 4 l + 6 of the chunk, the first
    //     four of which no later layer touches; they are written out after the layer and their ring slots zeroed;
    //   * a DoF that only this workgroup's patches touch (sa.flag[d] = 0 in all three directions) is stored once, with
    //     the vector update PreconditionChebyshev hands over (FdmFused) applied in the same instruction stream; a DoF
    //     shared with a neighbouring column / chunk or with the cells outside the box (3 DoF planes per side of a
    //     column) is added with a global atomic to a value that s7_init_kernel has set to the update's other terms.
    // HBM traffic: src once per patch row through the caches, dst once; no temporary.
    // ---------------------------------------------------------------------------------------------------------------
    constexpr int AY = 4 * AW + 3, AXP = 4 * PPW + 4, RING = 8; // rows, padded row length, planes of the accumulator

    // Writes finished planes of the accumulator.  A thread keeps its (at most EPT) row/column positions for the whole
    // march; per call it reads and zeroes all its entries, then issues every load of x / x_old, then stores: the
    // latencies of a layer's loads overlap instead of following each other entry by entry.
    constexpr int EPT = (AY * AXP + AW * 64 - 1) / (AW * 64); // accumulator entries of a plane per thread
    struct Emitter
    {
      const DevGrid &g;
      const FdmFused &fu;
      const Slice7Args &sa;
      double *       S_A, *dst;
      double         f2;
      int            oz, r[EPT];
      long long      oxy[EPT];
      int            kind[EPT]; // 0: not part of the column, 1: this workgroup's alone in x and y, 2: shared, 3: shared and constrained

      template <int NP>
      __device__ __forceinline__ void
      planes(const int P0) const
      {
        double    val[NP][EPT], xv[NP][EPT], xo[NP][EPT];
        long long o[NP][EPT];
        bool      own[NP][EPT];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          {
            const int       P = P0 + pl, sk = g.store_index_near(2, oz + P);
            double *        plane = S_A + (P & (RING - 1)) * (AY * AXP);
            const bool      fz = sa.flag[2][sk] != 0;
            const long long ok = sk * g.stride[2];
#pragma unroll
            for (int i = 0; i < EPT; ++i)
              {
                val[pl][i] = 0.;
                own[pl][i] = kind[i] == 1 && !fz;
                o[pl][i]   = oxy[i] + ok;
                if (kind[i])
                  {
                    val[pl][i]  = f2 * plane[r[i]];
                    plane[r[i]] = 0.;
                  }
              }
          }
        if (fu.mode == PC_STEP1 || fu.mode == PC_UPDATE)
          {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
              for (int i = 0; i < EPT; ++i)
                xv[pl][i] = own[pl][i] ? fu.x[o[pl][i]] : 0.;
          }
        if (fu.mode == PC_UPDATE)
          {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
              for (int i = 0; i < EPT; ++i)
                xo[pl][i] = own[pl][i] ? fu.x_old[o[pl][i]] : 0.;
          }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
          for (int i = 0; i < EPT; ++i)
            {
              if (own[pl][i])
                {
                  double t = val[pl][i];
                  if (fu.mode == PC_STEP1)
                    t = fma(fu.f1, xv[pl][i], t);
                  else if (fu.mode == PC_UPDATE)
                    t = xv[pl][i] + fu.f1 * (xv[pl][i] - xo[pl][i]) + t;
                  dst[o[pl][i]] = t;
                }
              else if (kind[i])
                {
                  // (a box patch never reaches a constrained DoF; the test keeps the contract with s7_init_kernel)
                  if (fu.mode == PC_PLAIN || !(kind[i] == 3 || g.dirichlet(2, oz + P0 + pl)))
                    unsafeAtomicAdd(&dst[o[pl][i]], val[pl][i]);
                }
            }
      }
    };

    template <bool HAS_PRE, bool HAS_POST, bool PERCELL>
    __global__ void __launch_bounds__(AW * 64, AW <= 4 ? 2 : 1)
    fdm_slice7_acc_kernel(const DevGrid g, const DevFdm f, const Slice7Args sa, const FdmFused fu, const double *__restrict__ src, double *dst)
    {
      extern __shared__ __attribute__((aligned(16))) double s7_lds[];
      double *S_T = s7_lds, *S_A = s7_lds + AW * TB7;
      const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
      const long long b  = xcd_block(blockIdx.x, gridDim.x);
      const int       tx = (int)(b % sa.ntx), ty = (int)((b / sa.ntx) % sa.nty), zc = (int)(b / ((long long)sa.ntx * sa.nty));
      const int cx0 = sa.lo[0] + tx * PPW, cy0 = sa.lo[1] + ty * AW, z_begin = sa.lo[2] + zc * sa.lz;
      const int z_end = min(z_begin + sa.lz, sa.lo[2] + sa.n[2]);
      const int ncx = min(PPW, sa.lo[0] + sa.n[0] - cx0), ncy = min(AW, sa.lo[1] + sa.n[1] - cy0); // cells of this column
      const int cols = 4 * ncx + 3, rows = 4 * ncy + 3;
      const int cl = lane / M, sub = lane - cl * M;
      const bool wave_on = w < ncy, active = wave_on && cl < ncx;
      const int  cx = cx0 + (cl < ncx ? cl : ncx - 1), cy = cy0 + (wave_on ? w : 0);
      for (int e = tid; e < RING * AY * AXP; e += AW * 64)
        S_A[e] = 0.;
      __syncthreads();
      // logical coordinates of the accumulator's first column / row / plane
      const int    ox = f.origin_mul * cx0 + f.origin_add, oy = f.origin_mul * cy0 + f.origin_add, oz = f.origin_mul * z_begin + f.origin_add;
      const double f2 = fu.mode == PC_PLAIN ? 1. : fu.f2;
      double *     acc = S_A + (4 * w) * AXP + 4 * cl + sub;
      Emitter      emit{g, fu, sa, S_A, dst, f2, oz};
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        {
          const int e = tid + i * (AW * 64), row = e / AXP, col = e - row * AXP;
          emit.r[i]    = e;
          emit.kind[i] = 0;
          emit.oxy[i]  = 0;
          if (row < rows && col < cols)
            {
              const int si = g.store_index_near(0, ox + col), sj = g.store_index_near(1, oy + row);
              emit.oxy[i]  = si * g.stride[0] + sj * g.stride[1];
              emit.kind[i] = !(sa.flag[0][si] | sa.flag[1][sj]) ? 1 : ((g.dirichlet(0, ox + col) || g.dirichlet(1, oy + row)) ? 3 : 2);
            }
        }
      for (int cz = z_begin; cz < z_end; ++cz)
        {
          const int lay = cz - z_begin;
          if (wave_on)
            {
              double v[M2];
              if (PERCELL)
                solve_percell<HAS_PRE, HAS_POST>(v, g, f, src, cx, cy, cz, lane, cl, sub, S_T + w * TB7);
              else
                solve_uniform<HAS_PRE, HAS_POST>(v, g, f, sa, src, cx, cy, cz, lane, cl, sub, S_T + w * TB7);
              if (active)
                {
#pragma unroll
                  for (int k = 0; k < M; ++k)
                    {
                      double *plane = acc + ((4 * lay + k) & (RING - 1)) * (AY * AXP);
#pragma unroll
                      for (int j = 0; j < M; ++j)
                        __hip_atomic_fetch_add(plane + j * AXP, v[k * M + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
          __syncthreads();
          emit.template planes<4>(4 * lay);
          if (cz == z_end - 1)
            emit.template planes<3>(4 * lay + 4);
          __syncthreads();
        }
    }

    // the terms of the fused update that do not come from the patches, on every DoF the accumulate kernel (or the generic
    // kernel, which takes the cells outside the box) adds to with atomics; constrained DoFs get their final value (unit
    // rows, include/matrix_free_internal.h:247-255) and are skipped by the atomics
    __global__ void __launch_bounds__(256)
    s7_init_kernel(const DevGrid g, const FdmFused fu, const unsigned char *__restrict__ f0, const unsigned char *__restrict__ f1,
                   const unsigned char *__restrict__ f2, const double *__restrict__ src, double *dst)
    {
      const int       sx = g.ext_hi[0] - g.ext_lo[0] + 1, sy = g.ext_hi[1] - g.ext_lo[1] + 1;
      const long long row = blockIdx.x; // (sj, sk)
      const int       sj = (int)(row % sy), sk = (int)(row / sy);
      const bool      fyz = f1[sj] | f2[sk];
      const bool      cyz = g.dirichlet(1, sj + g.ext_lo[1]) || g.dirichlet(2, sk + g.ext_lo[2]);
      for (int si = threadIdx.x; si < sx; si += blockDim.x)
        if (fyz | f0[si])
          {
            const long long o = si * g.stride[0] + sj * g.stride[1] + sk * g.stride[2];
            double          t = 0.;
            if (fu.mode == PC_STEP1)
              t = fu.f1 * fu.x[o];
            else if (fu.mode == PC_UPDATE)
              {
                const double xv = fu.x[o];
                t               = xv + fu.f1 * (xv - fu.x_old[o]);
              }
            if (fu.mode != PC_PLAIN && (cyz || g.dirichlet(0, si + g.ext_lo[0])))
              t += fu.f2 * src[o];
            dst[o] = t;
          }
    }
  } // namespace

  bool
  fdm_slice7(const DevGrid &g, const DevFdm &f, const double *src, double *tmp, hipStream_t st)
  {
    if (!f.s7_on || f.m != M)
      return false;
    Slice7Args sa;
    for (int d = 0; d < 3; ++d)
      {
        sa.lo[d] = f.s7_lo[d];
        sa.n[d]  = f.s7_hi[d] - f.s7_lo[d];
      }
    sa.npx = (sa.n[0] + PPW - 1) / PPW;
    sa.qx  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[0] : 0) * E;
    sa.qy  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[1] : 0) * E;
    sa.qz  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[2] : 0) * E;
    sa.inv = f.s7_inv;
    const long long n_waves = (long long)sa.npx * sa.n[1] * sa.n[2];
    const unsigned  blocks  = (unsigned)((n_waves + WPB - 1) / WPB);
    if (f.s7_on == 2)
      {
        // per-cell eigen-decompositions
        if (f.has_pre && f.has_post)
          hipLaunchKernelGGL((fdm_slice7_cell_kernel<true, true>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
        else if (f.has_pre)
          hipLaunchKernelGGL((fdm_slice7_cell_kernel<true, false>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
        else if (f.has_post)
          hipLaunchKernelGGL((fdm_slice7_cell_kernel<false, true>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
        else
          hipLaunchKernelGGL((fdm_slice7_cell_kernel<false, false>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
        return true;
      }
    if (f.has_pre && f.has_post)
      hipLaunchKernelGGL((fdm_slice7_kernel<true, true>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
    else if (f.has_pre)
      hipLaunchKernelGGL((fdm_slice7_kernel<true, false>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
    else if (f.has_post)
      hipLaunchKernelGGL((fdm_slice7_kernel<false, true>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
    else
      hipLaunchKernelGGL((fdm_slice7_kernel<false, false>), dim3(blocks), dim3(WPB * 64), 0, st, g, f, sa, src, tmp);
    return true;
  }

  bool
  fdm_slice7_acc(const DevGrid &g, const DevFdm &f, const double *src, double *dst, const FdmFused &fu_in, hipStream_t st)
  {
    if (!f.s7_on || f.m != M || f.origin_mul != 4 || f.s7_flag[0] == nullptr || f.s7_lz < 1)
      return false;
    FdmFused fu = fu_in;
    fu.direct   = 2;
    Slice7Args sa;
    for (int d = 0; d < 3; ++d)
      {
        sa.lo[d]   = f.s7_lo[d];
        sa.n[d]    = f.s7_hi[d] - f.s7_lo[d];
        sa.flag[d] = f.s7_flag[d];
      }
    sa.npx = (sa.n[0] + PPW - 1) / PPW;
    sa.qx  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[0] : 0) * E;
    sa.qy  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[1] : 0) * E;
    sa.qz  = f.eig + (long long)(f.s7_on == 1 ? f.s7_inst[2] : 0) * E;
    sa.inv = f.s7_inv;
    sa.ntx = sa.npx;
    sa.nty = (sa.n[1] + AW - 1) / AW;
    sa.lz  = f.s7_lz;
    sa.nzc = (sa.n[2] + sa.lz - 1) / sa.lz;
    const int sy = g.ext_hi[1] - g.ext_lo[1] + 1, sz = g.ext_hi[2] - g.ext_lo[2] + 1;
    hipLaunchKernelGGL(s7_init_kernel, dim3((unsigned)(sy * sz)), dim3(256), 0, st, g, fu, sa.flag[0], sa.flag[1], sa.flag[2], src, dst);
    if (!fdm_generic(g, f, src, dst, nullptr, fu, st)) // the cells outside the box
      return false;
    const unsigned blocks = (unsigned)(sa.ntx * sa.nty * sa.nzc);
    const size_t lds = sizeof(double) * (AW * TB7 + RING * AY * AXP);
    const auto   launch = [&](auto kernel) {
      if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
      hipLaunchKernelGGL(kernel, dim3(blocks), dim3(AW * 64), lds, st, g, f, sa, fu, src, dst);
      return true;
    };
    bool ok = false;
#define ASM_S7_ACC(PRE, POST) \
  ok = f.s7_on == 2 ? launch(fdm_slice7_acc_kernel<PRE, POST, true>) : launch(fdm_slice7_acc_kernel<PRE, POST, false>)
    if (f.has_pre && f.has_post)
      ASM_S7_ACC(true, true);
    else if (f.has_pre)
      ASM_S7_ACC(true, false);
    else if (f.has_post)
      ASM_S7_ACC(false, true);
    else
      ASM_S7_ACC(false, false);
#undef ASM_S7_ACC
    return ok;
  }
} // namespace asmamd
