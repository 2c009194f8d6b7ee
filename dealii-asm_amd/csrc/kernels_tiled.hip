// Tiled owner-computes sweeps for degree 4, cell patches with n_overlap = 1 (gfx950).
//
// Work decomposition
//   * a workgroup (8 waves = 512 threads) owns the DoFs of 11 x 7 cells in x-y and
//     marches along z over a chunk of cell layers; per layer it computes 12 x 8
//     cells (one halo column / row on the low sides, so that every owned DoF sees
//     all the cells that contribute to it);
//   * wave w handles cell row w of the tile: 12 cells x 5 lanes, lane (cell, k)
//     holds one 5 x 5 plane of the cell's 5 x 5 x 5 patch tile in registers;
//   * the two in-plane 1-D sweeps run on registers, the third direction is reached
//     by a 5-lane transpose through a small wave-private LDS buffer (no workgroup
//     barrier involved);
//   * the cells' output planes are summed into four LDS DoF planes (even cell rows,
//     barrier, odd cell rows, so that the order of the additions is fixed), the
//     z-neighbour's share travels from layer to layer through a wave-private LDS
//     carry, and the owner thread of every DoF applies the fused vector update and
//     stores it once, coalesced.
// This replaces, for the headline configuration, the gather / apply_inverse /
// scatter-add of ASPoissonPreconditioner::vmult_internal (include/matrix_free.h:
// 1023-1062), the cell loop of LaplaceOperatorMatrixFree::vmult (include/operator.h:
// 1367-1430) and the pre/post lambdas PreconditionChebyshev hands to both: dst needs
// no zeroing, no atomics on HBM and no compress(add); results are bitwise reproducible.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace asmamd
{
  bool
  tiled_kernels_available()
  {
    return true;
  }

  namespace
  {
    constexpr int P   = 4;
    constexpr int N1  = 5;
    constexpr int NE  = 25;       // values per cell plane
    constexpr int WX  = 12;       // cells per wave (x)
    constexpr int TY  = 8;        // cell rows per tile = waves per workgroup
    constexpr int OX  = WX - 1;   // owned cells per tile, x
    constexpr int OY  = TY - 1;   // owned cells per tile, y
    constexpr int NT  = TY * 64;  // threads
    constexpr int PX  = P * WX + 1; // DoFs per tile row (49)
    constexpr int PY  = P * TY + 1; // DoF rows per tile (33)
    constexpr int NR  = 5;        // rows per epilogue thread: Jr = wave + 8 r
    constexpr int EIG = NE + N1;  // eigen data per 1-D instance

    constexpr int LDS_OUT   = 4 * PY * PX;  // four DoF planes
    constexpr int TBUF      = 64 * N1;          // one transpose buffer (all 64 lanes write, no predicate)
    constexpr int LDS_T     = TY * 2 * TBUF;    // two per wave
    constexpr int LDS_CARRY = TY * WX * NE; // top planes of the previous layer
    constexpr int LDS_TAB   = 16 * EIG;     // eigen tables / reference matrices
    constexpr int LDS_TOTAL = LDS_OUT + LDS_T + LDS_CARRY + LDS_TAB;

    struct Tiling
    {
      int ntx, nty, nzc; // tiles in x, y; chunks in z
      int zlen;          // owned cell layers per chunk
    };

    // Workgroups b and b + 8 are observed to land on the same XCD (same L2).  Give the
    // workgroups of one XCD consecutive tiles, so that x-neighbours share their halo
    // columns and the partial cache lines at their common edge in one L2.  Placement
    // only affects speed, never results.
    __device__ __forceinline__ int
    xcd_remap(const int b, const int n)
    {
      const int xcd = b & 7, idx = b >> 3;
      return xcd * (n >> 3) + min(xcd, n & 7) + idx;
    }

    // is cell c (local coordinate, may be -1) computed by this rank?
    __device__ __forceinline__ bool
    cell_computable(const DevGrid &g, const int d, const int c)
    {
      if (c < -1 || c >= g.nc[d])
        return false;
      if (c >= 0)
        return true;
      return g.wrap[d] || g.ext_lo[d] < 0; // periodic image or halo cell of the lower rank
    }

    // is tile-relative DoF coordinate rel (logical I) written by this tile?
    __device__ __forceinline__ bool
    tile_owns(const DevGrid &g, const int d, const int rel, const int I, const int tile_cells)
    {
      if (I < 0 || I > g.own_hi[d] || rel < P)
        return false;
      return rel < P * tile_cells || I == g.nc[d] * P;
    }

    // FWD: out[a] = sum_i q[i][a] in[i];  else out[i] = sum_a q[i][a] in[a]
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const Q &q)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE == 1 || ASM_ABLATE >= 4)
      return; // diagnostic build: sweeps removed (results are wrong)
#endif
      double o[N1];
#pragma unroll
      for (int a = 0; a < N1; ++a)
        {
          double s = (FWD ? q[0 * N1 + a] : q[a * N1 + 0]) * u0;
          s        = fma(FWD ? q[1 * N1 + a] : q[a * N1 + 1], u1, s);
          s        = fma(FWD ? q[2 * N1 + a] : q[a * N1 + 2], u2, s);
          s        = fma(FWD ? q[3 * N1 + a] : q[a * N1 + 3], u3, s);
          s        = fma(FWD ? q[4 * N1 + a] : q[a * N1 + 4], u4, s);
          o[a]     = s;
        }
      u0 = o[0];
      u1 = o[1];
      u2 = o[2];
      u3 = o[3];
      u4 = o[4];
    }
    // sweep along the fast index of v[s*5 + f] / along the slow index
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_fast(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int s = 0; s < N1; ++s)
        line5<FWD>(v[s * 5 + 0], v[s * 5 + 1], v[s * 5 + 2], v[s * 5 + 3], v[s * 5 + 4], q);
    }
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_slow(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int f = 0; f < N1; ++f)
        line5<FWD>(v[0 * 5 + f], v[1 * 5 + f], v[2 * 5 + f], v[3 * 5 + f], v[4 * 5 + f], q);
    }

    // a wave-uniform 5x5 matrix in global memory: the loads become scalar loads and the
    // entries scalar operands of the FMAs, so they cost no vector registers
    struct UniformMatrix
    {
      // constant address space: uniform loads from it are scalar (s_load) loads
      const __attribute__((address_space(4))) double *p;
      __device__ __forceinline__ explicit UniformMatrix(const double *q)
        : p((const __attribute__((address_space(4))) double *)q)
      {}
      __device__ __forceinline__ double operator[](const int e) const { return p[e]; }
    };

    __device__ __forceinline__ void
    load25(double (&q)[NE], const double *__restrict__ src)
    {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        q[e] = src[e];
    }

    // ---------------------------------------------------------------------------
    // compute role of a lane
    // ---------------------------------------------------------------------------
    struct CellLane
    {
      int  lane, w, cl, sub; // sub = plane (layout A) or row (layout B) within the cell
      bool lane_ok;          // lanes 60..63 idle
      int  cx, cy;
      bool cell_xy;
      // storage offset of the cell's DoF (i, j) = off0 + j stride_y + i, except that the last
      // column / row may wrap around a periodic seam: xw4 / yw4 hold that correction (0 otherwise)
      int      off0, xw4, yw4;
      unsigned xd, yd; // Dirichlet masks of the cell's columns / rows
    };

    __device__ __forceinline__ void
    cell_lane_setup(const DevGrid &g, const int X0, const int Y0, CellLane &c)
    {
      c.lane    = threadIdx.x & 63;
      c.w       = threadIdx.x >> 6;
      c.cl      = c.lane / N1;
      c.sub     = c.lane - c.cl * N1;
      c.lane_ok = c.lane < WX * N1;
      c.cx      = X0 + c.cl;
      c.cy      = Y0 + c.w;
      c.cell_xy = c.lane_ok && cell_computable(g, 0, c.cx) && cell_computable(g, 1, c.cy);
      c.off0 = c.xw4 = c.yw4 = 0;
      c.xd = c.yd = 0u;
      if (c.cell_xy)
        {
          // every DoF of a computable cell exists; columns 0..3 / rows 0..3 are contiguous
          const int x0 = g.store_index_near(0, P * c.cx), y0 = g.store_index_near(1, P * c.cy);
          c.off0       = (int)(y0 * g.stride[1]) + x0;
          c.xw4        = g.store_index_near(0, P * c.cx + P) - (x0 + P);
          c.yw4        = (int)((g.store_index_near(1, P * c.cy + P) - (y0 + P)) * g.stride[1]);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            {
              c.xd |= (g.dirichlet(0, P * c.cx + i) ? 1u : 0u) << i;
              c.yd |= (g.dirichlet(1, P * c.cy + i) ? 1u : 0u) << i;
            }
        }
    }

    // gather plane K of the lane's cell; MASK: constrained DoFs read as zero.
    // Every DoF of a computable cell exists (the patch is the cell itself), so the 25
    // loads need no predicate; lanes without a cell read offset 0 and are zeroed by the
    // caller.  Dirichlet masking only touches lanes whose cell lies on the boundary.
    template <bool MASK>
    __device__ __forceinline__ void
    gather_plane(const DevGrid &g, const CellLane &c, const double *__restrict__ src, const int K, double (&v)[NE])
    {
      const double *pl = src + (c.cell_xy ? g.store_index_near(2, K) * g.stride[2] : 0) + c.off0;
#if defined(ASM_ABLATE) && ASM_ABLATE == 4
      for (int e = 0; e < NE; ++e)
        v[e] = 1.0 + e + K; // diagnostic build: no gather
      return;
#endif
      const int sy = c.cell_xy ? (int)g.stride[1] : 0;
#pragma unroll
      for (int j = 0; j < N1; ++j)
        {
          const double *row = pl + j * sy + (j == N1 - 1 ? c.yw4 : 0);
#pragma unroll
          for (int i = 0; i < N1 - 1; ++i)
            v[j * 5 + i] = row[i];
          v[j * 5 + N1 - 1] = row[N1 - 1 + c.xw4];
        }
      if (MASK)
        {
          const bool zd = g.dirichlet(2, K);
          if (c.cell_xy && (zd || (c.xd | c.yd) != 0u))
            {
#pragma unroll
              for (int j = 0; j < N1; ++j)
#pragma unroll
                for (int i = 0; i < N1; ++i)
                  if (zd || (((c.xd >> i) | (c.yd >> j)) & 1u))
                    v[j * 5 + i] = 0.;
            }
        }
    }

    // 5-lane transpose: lane `sub` of a cell holds v[s*5 + f]; afterwards it holds
    // v'[s*5 + f] = (value v[sub*5 + f] of the cell's lane s).  One fast index f per
    // round through the wave-private buffer (2 x 300 doubles).
    __device__ __forceinline__ void
    transpose5(double (&v)[NE], double *__restrict__ Tw, const CellLane &c)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE == 3 || ASM_ABLATE == 1 || ASM_ABLATE >= 4)
      return; // diagnostic build: transposes removed (results are wrong)
#endif
      double n[NE];
      // lanes 60..63 hold no cell: they run along unpredicated (their cl = 12 slots exist)
#pragma unroll
      for (int f = 0; f < N1; ++f)
        {
          double *buf = Tw + (f & 1) * TBUF;
#pragma unroll
          for (int s = 0; s < N1; ++s)
            buf[c.lane * N1 + s] = v[s * 5 + f];
          // the partner lanes' stores must be visible: LDS executes a wave's
          // instructions in order; keep the compiler from moving the loads up
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int s = 0; s < N1; ++s)
            n[s * 5 + f] = buf[(c.cl * N1 + s) * N1 + c.sub];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int e = 0; e < NE; ++e)
        v[e] = n[e];
    }

    // ---------------------------------------------------------------------------
    // epilogue role: thread (lane, wave) owns tile column Ir = lane (< 49) and rows
    // Jr = wave + 8 r, r = 0..4, of every output plane
    // ---------------------------------------------------------------------------
    struct PlaneRole
    {
      int       lane, wave;
      int       J0, ny, wrap_y, lo_y;
      long long stride_y;
      unsigned  row_owned, row_dirichlet;
      long long col_off;
      bool      col_in, col_owned, col_dirichlet;
    };

    __device__ __forceinline__ void
    role_setup(const DevGrid &g, const int I0, const int J0, PlaneRole &m)
    {
      m.lane     = threadIdx.x & 63;
      m.wave     = threadIdx.x >> 6;
      m.J0       = J0;
      m.ny       = g.nc[1] * g.p;
      m.wrap_y   = g.wrap[1];
      m.lo_y     = g.ext_lo[1];
      m.stride_y = g.stride[1];
      m.row_owned = m.row_dirichlet = 0u;
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int  Jr = m.wave + TY * r, J = J0 + Jr;
          const bool ow = Jr < PY && g.exists(1, J) && tile_owns(g, 1, Jr, J, TY);
          m.row_owned |= (ow ? 1u : 0u) << r;
          m.row_dirichlet |= ((ow && g.dirichlet(1, J)) ? 1u : 0u) << r;
        }
      const int I     = I0 + m.lane;
      m.col_in        = m.lane < PX;
      m.col_owned     = m.col_in && g.exists(0, I) && tile_owns(g, 0, m.lane, I, WX);
      m.col_off       = m.col_owned ? g.store_index_near(0, I) * g.stride[0] : 0;
      m.col_dirichlet = m.col_owned && g.dirichlet(0, I);
    }

    __device__ __forceinline__ long long
    row_offset(const PlaneRole &m, const int r)
    {
      int J = m.J0 + m.wave + TY * r;
      if (m.wrap_y)
        J += (J < 0) ? m.ny : ((J >= m.ny) ? -m.ny : 0);
      else
        J -= m.lo_y;
      return J * m.stride_y;
    }

    // operands of the fused update for the (up to) 20 DoFs a thread stores per layer;
    // loaded before the accumulation barriers so that their latency is hidden
    constexpr int NEP = (N1 - 1) * NR;
    template <int MODE>
    struct EpilogueOperands
    {
      static constexpr bool need_a = (MODE == OP_RESIDUAL || MODE == PC_STEP1 || MODE == PC_UPDATE);
      static constexpr bool need_b = (MODE == PC_UPDATE);
      double a[need_a ? NEP : 1];
      double b[need_b ? NEP : 1];
    };

    __device__ __forceinline__ long long
    dof_offset(const DevGrid &g, const PlaneRole &m, const int K, const int r)
    {
      return g.store_index_near(2, K) * g.stride[2] + m.col_off + row_offset(m, r);
    }

    template <int MODE>
    __device__ __forceinline__ void
    epilogue_prefetch(const DevGrid &g, const PlaneRole &m, const TiledArgs &a, const int K0, const int nplanes,
                      EpilogueOperands<MODE> &ops)
    {
      if (!EpilogueOperands<MODE>::need_a)
        return;
#if defined(ASM_ABLATE) && (ASM_ABLATE == 6 || ASM_ABLATE == 7)
      for (int e = 0; e < NEP; ++e)
        ops.a[e] = 1.0 + e; // diagnostic build: no operand loads
      return;
#endif
      const double *pa = (MODE == OP_RESIDUAL) ? a.b : a.x;
#pragma unroll
      for (int kk = 0; kk < N1 - 1; ++kk)
#pragma unroll
        for (int r = 0; r < NR; ++r)
          {
            const bool      own = kk < nplanes && m.col_owned && ((m.row_owned >> r) & 1u);
            const long long off = own ? dof_offset(g, m, K0 + kk, r) : 0;
            ops.a[kk * NR + r]  = pa[off];
            if (EpilogueOperands<MODE>::need_b)
              ops.b[kk * NR + r] = a.x_old[off];
          }
    }

    // store plane kk of S_out as DoF plane K (fused update of the mode) and clear it
    template <int MODE>
    __device__ __forceinline__ void
    plane_epilogue(const DevGrid &g, const PlaneRole &m, const TiledArgs &a, double *__restrict__ S_out, const int kk,
                   const int K, const EpilogueOperands<MODE> &ops)
    {
      if (!m.col_in)
        return;
      const bool dk = g.dirichlet(2, K) || m.col_dirichlet;
      double *   pl = S_out + kk * (PY * PX) + m.lane;
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int Jr = m.wave + TY * r;
          if (Jr < PY)
            {
              double t    = pl[Jr * PX];
              pl[Jr * PX] = 0.;
#if defined(ASM_ABLATE) && ASM_ABLATE == 6
              if (t != 1.2345e300)
                continue; // diagnostic build: no stores
#endif
              if (m.col_owned && ((m.row_owned >> r) & 1u))
                {
                  const long long off         = dof_offset(g, m, K, r);
                  const bool      constrained = dk || ((m.row_dirichlet >> r) & 1u);
                  const double    oa = ops.a[EpilogueOperands<MODE>::need_a ? kk * NR + r : 0];
                  const double    ob = ops.b[EpilogueOperands<MODE>::need_b ? kk * NR + r : 0];
                  if (MODE == OP_PLAIN || MODE == PC_PLAIN)
                    a.dst[off] = constrained ? 0. : t;
                  else if (MODE == OP_RESIDUAL)
                    // (A x)_c = x_c on constrained DoFs (matrix_free_internal.h:247-255)
                    a.dst[off] = oa - (constrained ? a.src[off] : t);
                  else
                    {
                      if (constrained)
                        t = a.src[off]; // unit rows of P^-1 on constrained DoFs
                      if (MODE == PC_FIRST)
                        a.dst[off] = a.f2 * t;
                      else if (MODE == PC_STEP1)
                        a.dst[off] = a.f1 * oa + a.f2 * t;
                      else
                        a.dst[off] = oa + a.f1 * (oa - ob) + a.f2 * t;
                    }
                }
            }
        }
    }

    // add the cell plane o (lane (cell, k), k < 4) into output plane k
    __device__ __forceinline__ void
    accumulate_plane(double *__restrict__ S_out, const CellLane &c, const int kk, const double (&o)[NE])
    {
      double *pl = S_out + kk * (PY * PX) + (P * c.w) * PX + P * c.cl;
#if defined(ASM_ABLATE) && ASM_ABLATE == 5
      if (o[3] == 1.2345e300)
        pl[0] = o[0]; // diagnostic build: no accumulation
      return;
#endif
#pragma unroll
      for (int j = 0; j < N1; ++j)
#pragma unroll
        for (int i = 0; i < N1; ++i)
          __hip_atomic_fetch_add(pl + j * PX + i, o[j * 5 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // carry exchange + accumulation of one layer's cell planes
    // On return o holds the gathered src plane of the NEXT layer (k_next >= 0): the loads are
    // issued as soon as o is dead, so they travel under the barriers and the store epilogue.
    template <int MODE, bool MASK>
    __device__ __forceinline__ void
    exchange_and_store(const DevGrid &g, const PlaneRole &m, const CellLane &c, const TiledArgs &a, double *__restrict__ S_out,
                       double *__restrict__ S_carry, double (&o)[NE], const bool active, const bool halo_layer, const int cz,
                       const int k_next)
    {
      EpilogueOperands<MODE> ops;
      if (!halo_layer)
        epilogue_prefetch<MODE>(g, m, a, P * cz, N1 - 1, ops);
      double *cw = S_carry + (c.w * WX + c.cl) * NE;
      // plane 0 receives the top plane of the layer below (same cell column)
      if (c.lane_ok && c.sub == 0)
        {
#pragma unroll
          for (int e = 0; e < NE; ++e)
            o[e] += cw[e];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c.lane_ok && c.sub == N1 - 1)
        {
#pragma unroll
          for (int e = 0; e < NE; ++e)
            cw[e] = active ? o[e] : 0.;
        }
      if (halo_layer)
        {
          // wave-uniform: the halo layer only feeds the carry
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (k_next >= 0)
            gather_plane<MASK>(g, c, a.src, k_next, o);
          return;
        }
#if defined(ASM_ABLATE) && ASM_ABLATE == 2
      // diagnostic build: no accumulation, no epilogue (results are wrong)
      if (active && o[0] == 1.2345e300)
        a.dst[0] = o[1];
      if (k_next >= 0)
        gather_plane<MASK>(g, c, a.src, k_next, o);
      return;
#endif
      // even cell rows, then odd cell rows: rows of equal parity share no DoF row
      if ((c.w & 1) == 0)
        {
          if (active && c.sub < N1 - 1)
            accumulate_plane(S_out, c, c.sub, o);
          if (k_next >= 0)
            gather_plane<MASK>(g, c, a.src, k_next, o);
        }
      __syncthreads();
      if ((c.w & 1) == 1)
        {
          if (active && c.sub < N1 - 1)
            accumulate_plane(S_out, c, c.sub, o);
          if (k_next >= 0)
            gather_plane<MASK>(g, c, a.src, k_next, o);
        }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < N1 - 1; ++kk)
        plane_epilogue<MODE>(g, m, a, S_out, kk, P * cz + kk, ops);
      __syncthreads();
    }

    // top plane of the block (physical boundary): only the last layer contributes
    template <int MODE>
    __device__ __forceinline__ void
    store_top_plane(const DevGrid &g, const PlaneRole &m, const CellLane &c, const TiledArgs &a, double *__restrict__ S_out,
                    double *__restrict__ S_carry)
    {
      EpilogueOperands<MODE> ops;
      epilogue_prefetch<MODE>(g, m, a, P * g.nc[2], 1, ops);
      double        o[NE];
      const double *cw = S_carry + (c.w * WX + c.cl) * NE;
      const bool    on = c.lane_ok && c.sub == 0;
      if (on)
        {
#pragma unroll
          for (int e = 0; e < NE; ++e)
            o[e] = cw[e];
        }
      if ((c.w & 1) == 0 && on)
        accumulate_plane(S_out, c, 0, o);
      __syncthreads();
      if ((c.w & 1) == 1 && on)
        accumulate_plane(S_out, c, 0, o);
      __syncthreads();
      plane_epilogue<MODE>(g, m, a, S_out, 0, P * g.nc[2], ops);
    }

    __device__ __forceinline__ void
    lds_clear(double *__restrict__ S_out, double *__restrict__ S_carry)
    {
      for (int e = threadIdx.x; e < LDS_OUT; e += NT)
        S_out[e] = 0.;
      for (int e = threadIdx.x; e < LDS_CARRY; e += NT)
        S_carry[e] = 0.;
    }

    // ---------------------------------------------------------------------------
    // FDM sweep
    // ---------------------------------------------------------------------------
    template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
    __global__ void __launch_bounds__(NT, 2)
    fdm_tiled_kernel(const DevGrid g, const DevFdm f, const TiledArgs a, const Tiling tl)
    {
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_out   = lds;
      double *S_T     = S_out + LDS_OUT;
      double *S_carry = S_T + LDS_T;
      double *S_tab   = S_carry + LDS_CARRY;

      const int b  = xcd_remap(blockIdx.x, gridDim.x);
      const int tx = b % tl.ntx, ty = (b / tl.ntx) % tl.nty, zc = b / (tl.ntx * tl.nty);
      const int X0 = tx * OX - 1, Y0 = ty * OY - 1; // tile cell (0,0)

      PlaneRole m;
      role_setup(g, P * X0, P * Y0, m);
      CellLane c;
      cell_lane_setup(g, X0, Y0, c);
      double *Tw = S_T + c.w * (2 * TBUF);

      // eigen tables of tensor-product meshes live in LDS (the launcher checks n_inst <= 16)
      if (TENSOR)
        for (int e = threadIdx.x; e < f.n_inst * EIG; e += NT)
          S_tab[e] = f.eig[e];
      lds_clear(S_out, S_carry);
      __syncthreads();

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;

      double v[NE]; // the lane's plane; reloaded with the next layer inside exchange_and_store
      gather_plane<false>(g, c, a.src, P * z_first + c.sub, v);
      const int cxs = c.cell_xy ? c.cx : 0, cys = c.cell_xy ? c.cy : 0; // table rows that exist
      for (int cz = z_first; cz < Ze; ++cz)
        {
          const bool active = c.cell_xy;
          int        ix = 0, iy = 0, iz = 0;
          if (active)
            {
              const long long ce =
                (c.cx + 1) + (long long)(g.nc[0] + 2) * ((c.cy + 1) + (long long)(g.nc[1] + 2) * (cz + 1));
              ix = TENSOR ? f.inst[0][c.cx + 1] : f.inst[0][ce];
              iy = TENSOR ? f.inst[1][c.cy + 1] : f.inst[1][ce];
              iz = TENSOR ? f.inst[2][cz + 1] : f.inst[2][ce];
            }
          // two address spaces, two pointer sets: a merged pointer would compile to flat loads
          const double *exl = S_tab + ix * EIG, *eyl = S_tab + iy * EIG, *ezl = S_tab + iz * EIG;
          const double *exg = f.eig + (long long)ix * EIG, *eyg = f.eig + (long long)iy * EIG,
                       *ezg = f.eig + (long long)iz * EIG;
          // tensor-product meshes: the y and z matrices depend on the cell row / layer only
          // (taken from the row / layer index, not from a lane: lane 0 may hold a cell that does not exist)
          const int iy_u = TENSOR ? f.inst[1][min(max(Y0 + (int)(threadIdx.x >> 6), -1), g.nc[1]) + 1] : 0;
          const int iz_u = TENSOR ? f.inst[2][cz + 1] : 0;
          const UniformMatrix Uy(f.eig + (long long)__builtin_amdgcn_readfirstlane(iy_u) * EIG),
            Uz(f.eig + (long long)__builtin_amdgcn_readfirstlane(iz_u) * EIG);
          // the x matrix is wave-uniform too unless the cell row touches a boundary column
          unsigned long long amask = __ballot(active);
          const int          lane0 = amask ? __ffsll((long long)amask) - 1 : 0;
          const int          ix_u  = __builtin_amdgcn_readlane(ix, lane0);
          const bool         x_uniform = TENSOR && __all(!active || ix == ix_u);
          const UniformMatrix Ux(f.eig + (long long)ix_u * EIG);
#define LOADQ(q, d) \
  do                \
    {               \
      if (TENSOR)   \
        load25(q, d##l); \
      else          \
        load25(q, d##g); \
    }               \
  while (0)

          double q[NE];
          // Lanes without a cell run along unpredicated on harmless data (table row 0, src
          // offset 0); nothing they compute is ever accumulated or carried.
          // ---- layout A: lane = (cell, z plane k); x and y forward sweeps -----------
          if (HAS_PRE)
            {
              const double wz = f.w_pre[2][(cz + 1) * N1 + c.sub];
#pragma unroll
              for (int j = 0; j < N1; ++j)
                {
                  const double wyz = f.w_pre[1][(cys + 1) * N1 + j] * wz;
#pragma unroll
                  for (int i = 0; i < N1; ++i)
                    v[j * 5 + i] *= f.w_pre[0][(cxs + 1) * N1 + i] * wyz;
                }
            }
          if (x_uniform)
            sweep_fast<true>(v, Ux);
          else
            {
              LOADQ(q, ex);
              sweep_fast<true>(v, q);
            }
          if (TENSOR)
            sweep_slow<true>(v, Uy);
          else
            {
              LOADQ(q, ey);
              sweep_slow<true>(v, q);
            }
          // ---- layout B: lane = (cell, y index b), v[c*5 + a]; z sweeps + scaling ----
          transpose5(v, Tw, c);
          if (TENSOR)
            sweep_slow<true>(v, Uz);
          else
            {
              LOADQ(q, ez);
              sweep_slow<true>(v, q);
            }
          if (f.inv_table != nullptr)
            {
              const double *inv = f.inv_table + (long long)((iz * f.n_inst + iy) * f.n_inst + ix) * (N1 * NE) + c.sub * N1;
#pragma unroll
              for (int cc = 0; cc < N1; ++cc)
#pragma unroll
                for (int aa = 0; aa < N1; ++aa)
                  v[cc * 5 + aa] *= inv[cc * NE + aa];
            }
          else
            {
              const double lyb = TENSOR ? eyl[NE + c.sub] : eyg[NE + c.sub];
#pragma unroll
              for (int cc = 0; cc < N1; ++cc)
#pragma unroll
                for (int aa = 0; aa < N1; ++aa)
                  {
                    // reciprocal with two Newton steps (full double precision)
                    const double sm = (TENSOR ? exl[NE + aa] : exg[NE + aa]) + (lyb + (TENSOR ? ezl[NE + cc] : ezg[NE + cc]));
                    double       r  = __builtin_amdgcn_rcp(sm);
                    r               = fma(fma(-sm, r, 1.0), r, r);
                    r               = fma(fma(-sm, r, 1.0), r, r);
                    v[cc * 5 + aa] *= r;
                  }
            }
          if (TENSOR)
            sweep_slow<false>(v, Uz);
          else
            sweep_slow<false>(v, q);
          // ---- back to layout A; y and x backward sweeps, post weights ---------------
          transpose5(v, Tw, c);
          if (TENSOR)
            sweep_slow<false>(v, Uy);
          else
            {
              LOADQ(q, ey);
              sweep_slow<false>(v, q);
            }
          if (x_uniform)
            sweep_fast<false>(v, Ux);
          else
            {
              LOADQ(q, ex);
              sweep_fast<false>(v, q);
            }
          if (HAS_POST)
            {
              const double wz = f.w_post[2][(cz + 1) * N1 + c.sub];
#pragma unroll
              for (int j = 0; j < N1; ++j)
                {
                  const double wyz = f.w_post[1][(cys + 1) * N1 + j] * wz;
#pragma unroll
                  for (int i = 0; i < N1; ++i)
                    v[j * 5 + i] *= f.w_post[0][(cxs + 1) * N1 + i] * wyz;
                }
            }
          exchange_and_store<MODE, false>(g, m, c, a, S_out, S_carry, v, active, cz < Zb, cz,
                                          cz + 1 < Ze ? P * (cz + 1) + c.sub : -1);
        }
#undef LOADQ
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        store_top_plane<MODE>(g, m, c, a, S_out, S_carry);
    }

    // ---------------------------------------------------------------------------
    // Laplace sweep on Cartesian cells.  For a box cell the sum-factorised
    // evaluate/integrate pair with exact Gauss quadrature (include/operator.h:866-875)
    // equals the Kronecker sum  hy hz/hx Mz My Kx + hx hz/hy Mz Ky Mx + hx hy/hz Kz My Mx
    // of the 1-D reference mass/stiffness matrices, which needs 7 one-dimensional
    // sweeps instead of 12 and no quadrature-point data.
    // ---------------------------------------------------------------------------
    template <int MODE>
    __global__ void __launch_bounds__(NT, 2)
    laplace_cart_tiled_kernel(const DevGrid g, const DevOp op, const TiledArgs a, const Tiling tl)
    {
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_out   = lds;
      double *S_T     = S_out + LDS_OUT;
      double *S_carry = S_T + LDS_T;
      double *S_tab   = S_carry + LDS_CARRY;

      const int b  = xcd_remap(blockIdx.x, gridDim.x);
      const int tx = b % tl.ntx, ty = (b / tl.ntx) % tl.nty, zc = b / (tl.ntx * tl.nty);
      const int X0 = tx * OX - 1, Y0 = ty * OY - 1;

      PlaneRole m;
      role_setup(g, P * X0, P * Y0, m);
      CellLane c;
      cell_lane_setup(g, X0, Y0, c);
      double *Tw = S_T + c.w * (2 * TBUF);

      // reference matrices (row-major [i][j], symmetric): wave-uniform scalar operands
      const UniformMatrix Mm(op.M_ref), Km(op.K_ref);
      (void)S_tab;
      lds_clear(S_out, S_carry);
      __syncthreads();

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;
      double    hx = 1., hy = 1.;
      if (c.cell_xy)
        {
          hx = op.h[0][c.cx + 1];
          hy = op.h[1][c.cy + 1];
        }

      double u[NE]; // the lane's plane; reloaded with the next layer inside exchange_and_store
      gather_plane<true>(g, c, a.src, P * z_first + c.sub, u);
      for (int cz = z_first; cz < Ze; ++cz)
        {
          const bool   active = c.cell_xy;
          const double hz     = op.h[2][cz + 1];
          double       t2[NE];
          // lanes without a cell run along unpredicated; nothing of theirs is accumulated or carried
          // ---- layout A: x and y sweeps ------------------------------------------------
#pragma unroll
          for (int e = 0; e < NE; ++e)
            t2[e] = u[e];
          sweep_fast<true>(t2, Km); // t2 = Kx u
          sweep_fast<true>(u, Mm);  // u  = Mx u
          const double alpha = hy * hz / hx, beta = hx * hz / hy, gamma = hx * hy / hz;
          // column by column: t2 <- alpha My t2 + beta Ky u  (needs Mz),  u <- gamma My u  (needs Kz)
#pragma unroll
          for (int f = 0; f < N1; ++f)
            {
              double k0 = u[0 * 5 + f], k1 = u[1 * 5 + f], k2 = u[2 * 5 + f], k3 = u[3 * 5 + f], k4 = u[4 * 5 + f];
              line5<true>(k0, k1, k2, k3, k4, Km); // Ky (Mx u)
              line5<true>(t2[0 * 5 + f], t2[1 * 5 + f], t2[2 * 5 + f], t2[3 * 5 + f], t2[4 * 5 + f], Mm);
              line5<true>(u[0 * 5 + f], u[1 * 5 + f], u[2 * 5 + f], u[3 * 5 + f], u[4 * 5 + f], Mm);
              t2[0 * 5 + f] = fma(beta, k0, alpha * t2[0 * 5 + f]);
              t2[1 * 5 + f] = fma(beta, k1, alpha * t2[1 * 5 + f]);
              t2[2 * 5 + f] = fma(beta, k2, alpha * t2[2 * 5 + f]);
              t2[3 * 5 + f] = fma(beta, k3, alpha * t2[3 * 5 + f]);
              t2[4 * 5 + f] = fma(beta, k4, alpha * t2[4 * 5 + f]);
              u[0 * 5 + f] *= gamma;
              u[1 * 5 + f] *= gamma;
              u[2 * 5 + f] *= gamma;
              u[3 * 5 + f] *= gamma;
              u[4 * 5 + f] *= gamma;
            }
          // ---- layout B: z sweeps ------------------------------------------------------
          transpose5(t2, Tw, c);
          transpose5(u, Tw, c);
          sweep_slow<true>(u, Km);  // Kz (My Mx u)
          sweep_slow<true>(t2, Mm); // Mz (...)
#pragma unroll
          for (int e = 0; e < NE; ++e)
            u[e] += t2[e];
          transpose5(u, Tw, c);
          exchange_and_store<MODE, true>(g, m, c, a, S_out, S_carry, u, active, cz < Zb, cz,
                                         cz + 1 < Ze ? P * (cz + 1) + c.sub : -1);
        }
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        store_top_plane<MODE>(g, m, c, a, S_out, S_carry);
    }

    Tiling
    make_tiling(const DevGrid &g)
    {
      Tiling tl;
      tl.ntx = (g.nc[0] + OX - 1) / OX;
      tl.nty = (g.nc[1] + OY - 1) / OY;
      // the last owned plane nc*p of a non-wrapped direction is produced by the tile
      // that holds cell nc-1, no extra tile needed
      const int tiles = tl.ntx * tl.nty;
      // one workgroup per CU: aim at ~2 x 256 workgroups, chunks of >= 4 layers
      int best = std::max(1, std::min(512 / tiles, g.nc[2] / 4));
      best     = std::min(best, g.nc[2]);
      tl.zlen = (g.nc[2] + best - 1) / best;
      tl.nzc  = (g.nc[2] + tl.zlen - 1) / tl.zlen;
      return tl;
    }

    template <typename K>
    void
    set_lds_limit(K kernel)
    {
      (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(LDS_TOTAL * sizeof(double)));
    }
  } // namespace

  template <int MODE>
  static void
  launch_laplace_tiled(const DevGrid &g, const DevOp &op, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    static bool done = false;
    if (!done)
      {
        set_lds_limit(laplace_cart_tiled_kernel<MODE>);
        done = true;
      }
    hipLaunchKernelGGL((laplace_cart_tiled_kernel<MODE>), dim3(tl.ntx * tl.nty * tl.nzc), dim3(NT), LDS_TOTAL * sizeof(double),
                       st, g, op, a, tl);
  }

  bool
  laplace_tiled(const DevGrid &g, const DevOp &op, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || !op.cartesian)
      return false;
    const Tiling tl = make_tiling(g);
    if (a.mode == OP_PLAIN)
      launch_laplace_tiled<OP_PLAIN>(g, op, a, tl, st);
    else if (a.mode == OP_RESIDUAL)
      launch_laplace_tiled<OP_RESIDUAL>(g, op, a, tl, st);
    else
      return false;
    return true;
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
  static void
  launch_fdm_tiled_t(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    static bool done = false;
    if (!done)
      {
        set_lds_limit(fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>);
        done = true;
      }
    hipLaunchKernelGGL((fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>), dim3(tl.ntx * tl.nty * tl.nzc), dim3(NT),
                       LDS_TOTAL * sizeof(double), st, g, f, a, tl);
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST>
  static void
  launch_fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    if (f.tensor_inst && f.n_inst <= 16)
      launch_fdm_tiled_t<MODE, HAS_PRE, HAS_POST, true>(g, f, a, tl, st);
    else
      launch_fdm_tiled_t<MODE, HAS_PRE, HAS_POST, false>(g, f, a, tl, st);
  }

  template <bool HAS_PRE, bool HAS_POST>
  static bool
  dispatch_fdm_mode(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    switch (a.mode)
      {
        case PC_PLAIN: launch_fdm_tiled<PC_PLAIN, HAS_PRE, HAS_POST>(g, f, a, tl, st); return true;
        case PC_FIRST: launch_fdm_tiled<PC_FIRST, HAS_PRE, HAS_POST>(g, f, a, tl, st); return true;
        case PC_STEP1: launch_fdm_tiled<PC_STEP1, HAS_PRE, HAS_POST>(g, f, a, tl, st); return true;
        case PC_UPDATE: launch_fdm_tiled<PC_UPDATE, HAS_PRE, HAS_POST>(g, f, a, tl, st); return true;
        default: return false;
      }
  }

  bool
  fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || f.m != N1 || f.vertex)
      return false;
    const Tiling tl = make_tiling(g);
    if (f.has_pre && f.has_post)
      return dispatch_fdm_mode<true, true>(g, f, a, tl, st);
    if (f.has_pre)
      return dispatch_fdm_mode<true, false>(g, f, a, tl, st);
    if (f.has_post)
      return dispatch_fdm_mode<false, true>(g, f, a, tl, st);
    return dispatch_fdm_mode<false, false>(g, f, a, tl, st);
  }
} // namespace asmamd
