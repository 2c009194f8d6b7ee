// Tiled owner-computes sweeps for degree 4, cell patches with n_overlap = 1 (gfx950).
//
// Work decomposition
//   * a workgroup (8 waves = 512 threads) owns the DoFs of 11 x 7 cells in x-y and
//     marches along z over a chunk of cell layers; per layer it computes 12 x 8
//     cells (one halo column / row on the low sides, so that every owned DoF sees
//     all the cells that contribute to it);
//   * wave w handles cell row w of the tile: 12 cells x 5 lanes, a lane holds one
//     5 x 5 slice of the cell's 5 x 5 x 5 patch tile in registers;
//   * the source rows of a layer (5 planes x 5 DoF rows x 49 DoFs per wave) are copied
//     global -> LDS by direct-to-LDS loads one layer ahead, in full lines, and picked
//     up from the wave-private image;
//   * two 1-D sweep directions run on registers, the third direction is reached
//     by a 5-lane transpose through a small wave-private LDS buffer (no workgroup
//     barrier involved); matrices whose columns are even / odd vectors are applied in
//     even-odd form (13 products per line) with wave-uniform scalar operands;
//   * contributions to shared DoFs are merged where they are produced: in x by a
//     5-lane shift inside the wave, in z through registers the lane keeps from one
//     layer to the next, in y through a 20-double LDS slot per cell (one barrier per
//     layer); the lane that then holds a DoF's final value applies the fused vector
//     update and stores it once (16-byte stores, rows of 4 DoFs).
// This replaces, for the headline configuration, the gather / apply_inverse /
// scatter-add of ASPoissonPreconditioner::vmult_internal (include/matrix_free.h:
// 1023-1062), the cell loop of LaplaceOperatorMatrixFree::vmult (include/operator.h:
// 1367-1430) and the pre/post lambdas PreconditionChebyshev hands to both: dst needs
// no zeroing, no atomics and no compress(add); results are bitwise reproducible.
// Diagnostic builds: -DASM_ABLATE=<bit mask> removes parts of the kernels (results wrong on
// purpose, tools/ablate.sh), -DASM_STAMP adds s_memtime stamps (tools/stamp_run.py).
#include "kernels.h"
#include "numerics.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstdlib>

namespace asmamd
{
#ifdef ASM_STAMP
  void tiled_debug_stamps(unsigned long long out[32], bool reset);
#endif

  namespace
  {
    constexpr int P   = 4;
    constexpr int N1  = 5;
    constexpr int NE  = 25;       // values per cell plane
    constexpr int WX  = 12;       // cells per wave (x)
#ifndef ASM_TY
#  define ASM_TY 8
#endif
    constexpr int TY  = ASM_TY;   // cell rows per tile = waves per workgroup
    constexpr int OX  = WX - 1;   // owned cells per tile, x
    constexpr int OY  = TY - 1;   // owned cells per tile, y
    constexpr int NT  = TY * 64;  // threads
    constexpr int EIG = NE + N1;  // eigen data per 1-D instance
    constexpr int PX_DOFS = P * WX + 1; // DoFs per tile row (49)
    constexpr int YS  = P * N1;   // doubles per y-merge slot: planes 0..3 x columns 0..4

    constexpr int TBUF      = 64 * N1;           // one transpose buffer (all 64 lanes write, no predicate)
    constexpr int YBUF      = WX * YS;           // y-merge slots of one wave and layer parity
    constexpr int LDS_T     = TY * TBUF;         // one per wave
    constexpr int LDS_Y     = 2 * TY * WX * YS;  // y-merge slots, double-buffered
    constexpr int LDS_TAB   = 16 * EIG;          // eigen tables
    constexpr int INV_MAX   = 3;                 // instances per direction up to which ...
    constexpr int LDS_INV   = INV_MAX * INV_MAX * N1 * NE; // ... the reciprocal eigenvalue sums of a layer fit
    // staged source rows of one wave and layer: 25 rows (5 planes x 5 DoF rows) of 25 16-byte chunks
    constexpr int GCH       = (PX_DOFS + 1) / 2; // chunks per row
    constexpr int GROW      = 2 * GCH;           // doubles per staged row (49 DoFs + 1)
    constexpr int GN        = (NE * GCH + 63) / 64; // wave instructions per layer (10)
    constexpr int GBUF      = GN * 128;          // doubles per wave
    constexpr int LDS_G     = TY * GBUF;
    constexpr int LDS_TOTAL = LDS_T + LDS_Y + LDS_G + LDS_TAB + LDS_INV;
    static_assert(LDS_TOTAL * sizeof(double) <= 160 * 1024, "one workgroup must fit the 160 KB LDS of a CU");

#ifdef ASM_STAMP
    // diagnostic build: cycles (s_memtime) an interior wave spends between the STAMP marks
    __device__ unsigned long long g_stamps[32];
    // ... and the life of every workgroup: {start, end (s_memrealtime, 100 MHz), XCC id << 32 | hardware id, kernel tag << 32 | block}
    __device__ unsigned long long g_wgtrace[16384][4];
    __device__ unsigned int       g_wgtrace_n;
#  define WGTRACE_BEGIN(tag)                                                                   \
    const unsigned long long wgt_t0 = __builtin_amdgcn_s_memrealtime();                       \
    constexpr unsigned       wgt_tag = tag;
#  define WGTRACE_END()                                                                        \
    do                                                                                        \
      {                                                                                       \
        __syncthreads();                                                                      \
        if (threadIdx.x == 0)                                                                 \
          {                                                                                   \
            const unsigned slot = atomicAdd(&g_wgtrace_n, 1u);                                \
            if (slot < 16384)                                                                 \
              {                                                                               \
                g_wgtrace[slot][0] = wgt_t0;                                                  \
                g_wgtrace[slot][1] = __builtin_amdgcn_s_memrealtime();                        \
                g_wgtrace[slot][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | \
                                     (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
                g_wgtrace[slot][3] = ((unsigned long long)wgt_tag << 32) | blockIdx.x;        \
              }                                                                               \
          }                                                                                   \
      }                                                                                       \
    while (0)
    struct StampCtx
    {
      bool               on;
      unsigned long long t, acc[16];
    };
#  define STAMP_DECL(base)                                                                  \
    StampCtx stamp_ctx;                                                                     \
    stamp_ctx.on = (tx == tl.ntx / 2 && ty == tl.nty / 2 && zc == tl.nzc / 2) && ((threadIdx.x >> 6) == 3); /* an interior tile */ \
    stamp_ctx.t  = __builtin_amdgcn_s_memtime();                                            \
    for (int i = 0; i < 16; ++i)                                                            \
      stamp_ctx.acc[i] = 0;                                                                 \
    constexpr int stamp_base = base;
#  define STAMP(i)                                                        \
    do                                                                    \
      {                                                                   \
        if (stamp_ctx.on)                                                 \
          {                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
            const unsigned long long t = __builtin_amdgcn_s_memtime();    \
            stamp_ctx.acc[i] += t - stamp_ctx.t;                          \
            stamp_ctx.t = t;                                              \
          }                                                               \
      }                                                                   \
    while (0)
#  define STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#  define STAMP_FLUSH()                                                          \
    do                                                                           \
      {                                                                          \
        if (stamp_ctx.on && (threadIdx.x & 63) == 0)                             \
          for (int i = 0; i < 16; ++i)                                           \
            atomicAdd(&g_stamps[stamp_base + i], stamp_ctx.acc[i]);              \
      }                                                                          \
    while (0)
#  define STAMP_PARAM , StampCtx &stamp_ctx
#  define STAMP_ARG , stamp_ctx
#else
#  define STAMP_DECL(base)
#  define STAMP(i)
#  define STAMP_WAIT_VM()
#  define WGTRACE_BEGIN(tag)
#  define WGTRACE_END()
#  define STAMP_FLUSH()
#  define STAMP_PARAM
#  define STAMP_ARG
#endif

    struct Tiling
    {
      int ntx, nty, nzc; // tiles in x, y; chunks in z
      int zlen;          // owned cell layers per chunk
      // first cell of tile 0 in x / y: -1 where the block has a cell below its first one (periodic image, halo cell of
      // the lower rank: tile 0 needs it as its halo column / row), 0 at a physical boundary - tile 0 then owns all its
      // twelve columns / TY rows, which saves a whole column (row) of tiles when nc = 1 (mod owned cells per tile),
      // e.g. 78 = 7 x 11 + 1 = 11 x 7 + 1: 7 x 11 instead of 8 x 12 tiles
      int x_first, y_first;
      // Two-phase launches (TiledArgs::phase 1 / 2): the workgroups in "phase order": the boundary ones (up to six
      // slabs around the box of interior workgroups) first, then the interior box, each box in lexicographic order.
      // Phase 1 is the first n_first of them (all boundary workgroups, padded with interior ones to whole rounds of
      // compute units), phase 2 the others; each phase is a dense launch, so that its workgroups spread evenly over
      // the XCDs whatever the shape of the boundary set.
      int n_box, box_lo[7][3], box_size[7][3], box_end[7]; // box_end: cumulative workgroup counts
      int n_first;
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

    // workgroup (tile x, tile y, z chunk) of this block
    __device__ __forceinline__ void
    workgroup_of_block(const TiledArgs &a, const Tiling &tl, int &tx, int &ty, int &zc)
    {
      int i = xcd_remap(blockIdx.x, gridDim.x);
      if (a.phase == 0)
        {
          if (a.order != nullptr) // expensive workgroups first (tiled_order); a scalar load
            i = ((const __attribute__((address_space(4))) int *)a.order)[i];
          tx = i % tl.ntx;
          ty = (i / tl.ntx) % tl.nty;
          zc = i / (tl.ntx * tl.nty);
          return;
        }
      int o = i + (a.phase == 2 ? tl.n_first : 0), k = 0, begin = 0;
      while (k + 1 < tl.n_box && o >= tl.box_end[k])
        begin = tl.box_end[k++];
      o -= begin;
      tx = tl.box_lo[k][0] + o % tl.box_size[k][0];
      ty = tl.box_lo[k][1] + (o / tl.box_size[k][0]) % tl.box_size[k][1];
      zc = tl.box_lo[k][2] + o / (tl.box_size[k][0] * tl.box_size[k][1]);
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

    // FWD: out[a] = sum_i q[i][a] in[i];  else out[i] = sum_a q[i][a] in[a]
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const Q &q)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE & 32)
      return; // diagnostic build (results are wrong)
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

    // ---- even-odd form (13 products + 4 sums per line instead of 25 products) for a matrix with
    // columns 0..2 even and columns 3, 4 odd vectors (parity-sorted eigenvectors, numerics.h:
    // sort_eigenvectors_by_parity); same meaning of FWD as line5
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    line5_eo(double &u0, double &u1, double &u2, double &u3, double &u4, const Q &q)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE & 32)
      return; // diagnostic build (results are wrong)
#endif
      if (FWD)
        {
          const double e0 = u0 + u4, e1 = u1 + u3, o0 = u0 - u4, o1 = u1 - u3, m = u2;
          u0 = fma(q[2 * N1 + 0], m, fma(q[1 * N1 + 0], e1, q[0 * N1 + 0] * e0));
          u1 = fma(q[2 * N1 + 1], m, fma(q[1 * N1 + 1], e1, q[0 * N1 + 1] * e0));
          u2 = fma(q[2 * N1 + 2], m, fma(q[1 * N1 + 2], e1, q[0 * N1 + 2] * e0));
          u3 = fma(q[1 * N1 + 3], o1, q[0 * N1 + 3] * o0);
          u4 = fma(q[1 * N1 + 4], o1, q[0 * N1 + 4] * o0);
        }
      else
        {
          const double E0 = fma(q[0 * N1 + 2], u2, fma(q[0 * N1 + 1], u1, q[0 * N1 + 0] * u0));
          const double E1 = fma(q[1 * N1 + 2], u2, fma(q[1 * N1 + 1], u1, q[1 * N1 + 0] * u0));
          const double E2 = fma(q[2 * N1 + 2], u2, fma(q[2 * N1 + 1], u1, q[2 * N1 + 0] * u0));
          const double O0 = fma(q[0 * N1 + 4], u4, q[0 * N1 + 3] * u3);
          const double O1 = fma(q[1 * N1 + 4], u4, q[1 * N1 + 3] * u3);
          u0              = E0 + O0;
          u4              = E0 - O0;
          u1              = E1 + O1;
          u3              = E1 - O1;
          u2              = E2;
        }
    }
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_fast_eo(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int s = 0; s < N1; ++s)
        line5_eo<FWD>(v[s * 5 + 0], v[s * 5 + 1], v[s * 5 + 2], v[s * 5 + 3], v[s * 5 + 4], q);
    }
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_slow_eo(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int f = 0; f < N1; ++f)
        line5_eo<FWD>(v[0 * 5 + f], v[1 * 5 + f], v[2 * 5 + f], v[3 * 5 + f], v[4 * 5 + f], q);
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
      // the same matrix behind a pointer the compiler cannot trace: its loads are issued where the
      // copy is made instead of once before the layer loop (three resident matrices do not fit into
      // the scalar registers and would be spilled to vector lanes)
      __device__ __forceinline__ UniformMatrix fresh() const
      {
        UniformMatrix r(*this);
        asm volatile("" : "+s"(r.p));
        return r;
      }
    };

    // read-only table entry at a wave-uniform index: a scalar load (s_load), so that it does not
    // enter the vector memory counter the staged rows and update operands are tracked with
    template <typename T>
    __device__ __forceinline__ T
    uniform_load(const T *table, const int idx)
    {
      return ((const __attribute__((address_space(4))) T *)table)[__builtin_amdgcn_readfirstlane(idx)];
    }

    __device__ __forceinline__ void
    load25(double (&q)[NE], const double *__restrict__ src)
    {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        q[e] = src[e];
    }

    // ---------------------------------------------------------------------------
    // role of a lane.  Two register layouts of a cell's 5 x 5 x 5 tile over its 5 lanes:
    //   layout A: lane sub = z plane k, registers v[j*5 + i]
    //   layout B: lane sub = y row   j, registers v[k*5 + i]
    // transpose5 switches between them.  Results are merged and stored in layout B.
    // ---------------------------------------------------------------------------
    struct CellLane
    {
      int  lane, w, cl, sub;
      bool lane_ok; // lanes 60..63 idle
      int  cx, cy;
      bool cell_xy;
      // storage offset of the cell's DoF (i, j) = off0 + j sy + i, except that the last
      // column / row may wrap around a periodic seam: xw4 / yw4 hold that correction (0 otherwise)
      int      off0, sy, xw4, yw4;
      int      rowoff; // off0 + sub sy (+ yw4): the lane's row in layout B
      unsigned xd, yd; // Dirichlet masks of the cell's columns / rows
      // ownership (layout B): the lane stores columns 0..3 (and 4 if own_c4) of its row
      bool own_row, own_c4;
      bool left_ok, below_ok; // the x / y neighbour cell whose last column / row ends in this cell is computed
      int  shl5;              // ds_bpermute address of lane - 5
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
      c.off0 = c.sy = c.xw4 = c.yw4 = c.rowoff = 0;
      c.xd = c.yd = 0u;
      if (c.cell_xy)
        {
          // every DoF of a computable cell exists; columns 0..3 / rows 0..3 are contiguous
          const int x0 = g.store_index_near(0, P * c.cx), y0 = g.store_index_near(1, P * c.cy);
          c.sy         = (int)g.stride[1];
          c.off0       = y0 * c.sy + x0;
          c.xw4        = g.store_index_near(0, P * c.cx + P) - (x0 + P);
          c.yw4        = (g.store_index_near(1, P * c.cy + P) - (y0 + P)) * c.sy;
          c.rowoff     = c.off0 + c.sub * c.sy + (c.sub == N1 - 1 ? c.yw4 : 0);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            {
              c.xd |= (g.dirichlet(0, P * c.cx + i) ? 1u : 0u) << i;
              c.yd |= (g.dirichlet(1, P * c.cy + i) ? 1u : 0u) << i;
            }
        }
      // a cell is stored by the tile that holds it outside its halo row / column; the cell's
      // DoFs (0..3)^3 belong to it, the block's last plane to the last cell
      // (the first column / row of a tile is its halo - unless nothing lies below it: Tiling::x_first, y_first)
      const bool owned_cell = c.cell_xy && (c.cl >= 1 || (c.cx >= 0 && !cell_computable(g, 0, c.cx - 1))) &&
                              (c.w >= 1 || (c.cy >= 0 && !cell_computable(g, 1, c.cy - 1)));
      const bool last_col   = !g.wrap[0] && g.own_hi[0] == g.nc[0] * P && c.cx == g.nc[0] - 1;
      const bool last_row   = !g.wrap[1] && g.own_hi[1] == g.nc[1] * P && c.cy == g.nc[1] - 1;
      c.own_row             = owned_cell && (c.sub < N1 - 1 || last_row);
      c.own_c4              = c.own_row && last_col;
      c.left_ok             = c.cell_xy && c.cl >= 1 && cell_computable(g, 0, c.cx - 1);
      c.below_ok            = c.cell_xy && c.w >= 1 && cell_computable(g, 1, c.cy - 1);
      c.shl5                = (c.lane >= N1 ? c.lane - N1 : c.lane) << 2;
    }

    __device__ __forceinline__ long long
    plane_offset(const DevGrid &g, const int K)
    {
      return g.store_index_near(2, K) * g.stride[2];
    }

    // storage offsets of the 5 DoF planes P cz .. P cz + 4 of cell layer cz (wave-uniform): one
    // wrap decision per layer for plane 0 (planes 1..3 follow it) and one for plane 4
    struct LayerPlanes
    {
      long long z[N1];
    };
    __device__ __forceinline__ LayerPlanes
    layer_planes(const DevGrid &g, const int cz)
    {
      LayerPlanes     lp;
      const long long s0 = g.store_index_near(2, P * cz);
#pragma unroll
      for (int k = 0; k < P; ++k)
        lp.z[k] = (s0 + k) * g.stride[2];
      lp.z[P] = plane_offset(g, P * cz + P);
      return lp;
    }

    // ---------------------------------------------------------------------------
    // Source staging.  A wave needs, per layer, 5 planes x 5 DoF rows of 49 consecutive DoFs.
    // They are copied global -> LDS by 10 direct-to-LDS loads (every lane one 16-byte chunk,
    // rows contiguous in LDS), issued one layer ahead: no vector registers are tied up while
    // the data is in flight, every row is read from memory in full lines and exactly once
    // per wave.  A lane then picks its 25 values from the wave-private image.
    // ---------------------------------------------------------------------------
    // rows touched by wave instruction n: planes kmin(n) .. kmax(n)
    __host__ __device__ constexpr int
    gather_kmin(const int n)
    {
      return (64 * n / GCH) / N1;
    }
    __host__ __device__ constexpr int
    gather_kmax(const int n)
    {
      return ((64 * n + 63) / GCH < NE ? (64 * n + 63) / GCH : NE - 1) / N1;
    }
    // Planes 0..3 of a cell layer are consecutive in storage; plane 4 may lie on the other side of a
    // periodic seam.  Exactly one instruction (GSEAM) covers rows of plane 3 and plane 4.
    constexpr int GSEAM = 7;
    static_assert(GN == 10 && gather_kmin(GSEAM) == P - 1 && gather_kmax(GSEAM) == P, "row -> instruction map changed");
    static_assert(gather_kmax(GSEAM - 1) == P - 1 && gather_kmin(GSEAM + 1) == P, "row -> instruction map changed");

    struct GatherPlan
    {
      // byte offset of the lane's chunk of instruction n relative to plane kmin(n) of the layer
      // (n != GSEAM: including the step to the next plane where the chunk lies in it)
      unsigned off8[GN];
      bool     seam_hi; // instruction GSEAM: the lane's chunk lies in plane 4
    };

    __device__ __forceinline__ void
    gather_plan_setup(const DevGrid &g, const int X0, const int Y0, GatherPlan &gp)
    {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      gp.seam_hi     = false;
#pragma unroll
      for (int n = 0; n < GN; ++n)
        {
          const int q = 64 * n + lane;
          int       r = q / GCH, cc = q - r * GCH;
          if (r >= NE) // tail of the last instruction: any valid address
            {
              r  = NE - 1;
              cc = 0;
            }
          const int k = r / N1, j = r - k * N1;
          const int I = P * X0 + 2 * cc, J = P * (Y0 + w) + j;
          // chunks outside the stored range belong to cells that are not computed: any valid address
          const int       xs = g.exists(0, I) ? g.store_index(0, I) : 0;
          const int       ys = g.exists(1, J) ? g.store_index(1, J) : 0;
          const long long o  = (long long)ys * g.stride[1] + xs + ((n != GSEAM && k > gather_kmin(n)) ? g.stride[2] : 0);
          gp.off8[n]         = (unsigned)(o * 8);
          if (n == GSEAM && k > gather_kmin(n))
            gp.seam_hi = true;
        }
    }

    // request the rows of cell layer cz (all GN direct-to-LDS loads of the wave)
    struct GatherRequest
    {
      const GatherPlan &gp;
      const double *    base[N1]; // src + storage offset of plane k (wave-uniform)
      unsigned          lds_base; // LDS byte address of the wave's image (wave-uniform)
      bool              on;

      __device__ __forceinline__ GatherRequest(const DevGrid &g, const GatherPlan &plan, const double *s, const int cz,
                                               double *Gw, const bool enabled)
        : gp(plan)
        , on(enabled)
      {
        lds_base =
          __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)Gw);
#pragma unroll
        for (int k = 0; k < N1; ++k)
          base[k] = s;
        if (enabled)
          {
            const LayerPlanes lp = layer_planes(g, cz);
#pragma unroll
            for (int k = 0; k < N1; ++k)
              base[k] = s + lp.z[k];
          }
      }

      __device__ __forceinline__ void
      issue() const
      {
#if defined(ASM_ABLATE) && (ASM_ABLATE & 1)
        return;
#endif
        if (!on)
          return;
        // the seam instruction: both planes relative to the lower of the two bases (offsets are unsigned)
        const double * b_seam = base[P] < base[P - 1] ? base[P] : base[P - 1];
        const unsigned d_lo = (unsigned)((base[P - 1] - b_seam) * 8), d_hi = (unsigned)((base[P] - b_seam) * 8);
        const unsigned o_seam = gp.off8[GSEAM] + (gp.seam_hi ? d_hi : d_lo);
        // One asm statement: hipcc would otherwise drain every pending vector memory operation
        // (vmcnt(0)) at the next LDS access that might alias the destination, i.e. at the transposes
        // right behind this request; the asm form is invisible to the compiler's wait-count
        // bookkeeping and gather_wait() is the explicit wait that covers it.  Scalar base + 32-bit
        // lane offset addressing; M0 = wave-uniform LDS destination of each 1 KB piece, restored at the end.
        unsigned keep;
        asm volatile("s_mov_b32 %[keep], m0\n\t"
                     "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b0]\n\t"
                     "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b0]\n\t"
                     "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b1]\n\t"
                     "s_add_u32 m0, %[lds], 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b1]\n\t"
                     "s_add_u32 m0, %[lds], 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b2]\n\t"
                     "s_add_u32 m0, %[lds], 0x1400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[b2]\n\t"
                     "s_add_u32 m0, %[lds], 0x1800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v6], %[b3]\n\t"
                     "s_add_u32 m0, %[lds], 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v7], %[bs]\n\t"
                     "s_add_u32 m0, %[lds], 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v8], %[b4]\n\t"
                     "s_add_u32 m0, %[lds], 0x2400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v9], %[b4]\n\t"
                     "s_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep)
                     : [lds] "s"(lds_base), [b0] "s"(base[0]), [b1] "s"(base[1]), [b2] "s"(base[2]), [b3] "s"(base[3]),
                       [b4] "s"(base[4]), [bs] "s"(b_seam), [v0] "v"(gp.off8[0]), [v1] "v"(gp.off8[1]), [v2] "v"(gp.off8[2]),
                       [v3] "v"(gp.off8[3]), [v4] "v"(gp.off8[4]), [v5] "v"(gp.off8[5]), [v6] "v"(gp.off8[6]), [v7] "v"(o_seam),
                       [v8] "v"(gp.off8[8]), [v9] "v"(gp.off8[9])
                     : "memory", "scc");
      }
    };
    static_assert(gather_kmin(0) == 0 && gather_kmin(1) == 0 && gather_kmin(2) == 1 && gather_kmin(3) == 1 && gather_kmin(4) == 2 &&
                    gather_kmin(5) == 2 && gather_kmin(6) == 3 && gather_kmin(8) == 4 && gather_kmin(9) == 4,
                  "the base operands of GatherRequest::issue follow gather_kmin");
    __device__ __forceinline__ void
    gather_issue(const DevGrid &g, const GatherPlan &gp, const double *__restrict__ src, const int cz, double *Gw)
    {
      const GatherRequest rq(g, gp, src, cz, Gw, true);
      rq.issue();
    }

    // the staged image is complete and may be read / has been read and may be overwritten
    __device__ __forceinline__ void
    gather_wait()
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ void
    gather_release()
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // pick the lane's 25 values: LAYOUT_B ? v[k*5 + i] of row sub : v[j*5 + i] of plane sub.
    // MASK: constrained DoFs read as zero (cz: the layer, for the z boundary planes)
    template <bool LAYOUT_B, bool MASK>
    __device__ __forceinline__ void
    staged_read(const DevGrid &g, const CellLane &c, const double *Gw, const int cz, double (&v)[NE])
    {
      const double *base = Gw + (LAYOUT_B ? c.sub : c.sub * N1) * GROW + P * c.cl;
#pragma unroll
      for (int s = 0; s < N1; ++s)
        {
          const double *row = base + s * (LAYOUT_B ? N1 * GROW : GROW);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            v[s * 5 + i] = row[i];
        }
      if (MASK)
        {
          // Dirichlet masks of the slow register index (bits) and of the lane's own slice
          const unsigned zd   = (g.dirichlet(2, P * cz) ? 1u : 0u) | (g.dirichlet(2, P * cz + P) ? 16u : 0u);
          const unsigned slow = LAYOUT_B ? zd : c.yd;
          const bool     mine = ((LAYOUT_B ? c.yd : zd) >> c.sub) & 1u;
          if (c.cell_xy && (mine || slow != 0u || c.xd != 0u))
            {
#pragma unroll
              for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int i = 0; i < N1; ++i)
                  if (mine || (((c.xd >> i) | (slow >> s)) & 1u))
                    v[s * 5 + i] = 0.;
            }
        }
    }


    // ---------------------------------------------------------------------------
    // Source staging, second form: only planes 1..4 of a cell layer are copied global -> LDS (8
    // direct-to-LDS loads per wave and layer instead of 10, 8 KB instead of 10 KB per wave); plane 0
    // is plane 4 of the layer below, which the lane still holds in registers (`top`).  Rows
    // r = 0..19 <-> (plane 1 + r / 5, DoF row r % 5).
    // ---------------------------------------------------------------------------
    constexpr int G4ROWS = P * N1;                      // 20 staged rows
    constexpr int G4N    = (G4ROWS * GCH + 63) / 64;    // wave instructions per layer (8)
    constexpr int G4BUF  = G4N * 128;                   // doubles per wave
    __host__ __device__ constexpr int
    gather4_kmin(const int n)
    {
      return 1 + (64 * n / GCH) / N1;
    }
    __host__ __device__ constexpr int
    gather4_kmax(const int n)
    {
      return 1 + ((64 * n + 63) / GCH < G4ROWS ? (64 * n + 63) / GCH : G4ROWS - 1) / N1;
    }
    constexpr int G4SEAM = 5; // the instruction that covers rows of plane 3 and of plane 4
    static_assert(G4N == 8 && gather4_kmin(G4SEAM) == P - 1 && gather4_kmax(G4SEAM) == P, "row -> instruction map changed");
    static_assert(gather4_kmax(G4SEAM - 1) == P - 1 && gather4_kmin(G4SEAM + 1) == P, "row -> instruction map changed");
    static_assert(gather4_kmin(0) == 1 && gather4_kmin(1) == 1 && gather4_kmin(2) == 2 && gather4_kmin(3) == 2 && gather4_kmin(4) == 3 &&
                    gather4_kmin(6) == 4 && gather4_kmin(7) == 4,
                  "the base operands of gather4_issue follow gather4_kmin");

    struct GatherPlan4
    {
      unsigned off8[G4N];
      bool     seam_hi;
    };

    __device__ __forceinline__ void
    gather4_plan_setup(const DevGrid &g, const int X0, const int Y0, GatherPlan4 &gp)
    {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      gp.seam_hi     = false;
#pragma unroll
      for (int n = 0; n < G4N; ++n)
        {
          const int q = 64 * n + lane;
          int       r = q / GCH, cc = q - r * GCH;
          if (r >= G4ROWS) // tail of the last instruction: any valid address
            {
              r  = G4ROWS - 1;
              cc = 0;
            }
          const int k = 1 + r / N1, j = r % N1;
          const int I = P * X0 + 2 * cc, J = P * (Y0 + w) + j;
          const int       xs = g.exists(0, I) ? g.store_index(0, I) : 0;
          const int       ys = g.exists(1, J) ? g.store_index(1, J) : 0;
          const long long o  = (long long)ys * g.stride[1] + xs + ((n != G4SEAM && k > gather4_kmin(n)) ? g.stride[2] : 0);
          gp.off8[n]         = (unsigned)(o * 8);
          if (n == G4SEAM && k > gather4_kmin(n))
            gp.seam_hi = true;
        }
    }

    // request planes 1..4 of cell layer cz (all G4N direct-to-LDS loads of the wave); see GatherRequest::issue
    __device__ __forceinline__ void
    gather4_issue(const DevGrid &g, const GatherPlan4 &gp, const double *__restrict__ s, const int cz, double *Gw, const bool on)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE & 1)
      return;
#endif
      if (!on)
        return;
      const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) double *)Gw);
      const LayerPlanes lp = layer_planes(g, cz);
      const double *    b1 = s + lp.z[1], *b2 = s + lp.z[2], *b3 = s + lp.z[3], *b4 = s + lp.z[4];
      const double *    b_seam = b4 < b3 ? b4 : b3;
      const unsigned    d_lo = (unsigned)((b3 - b_seam) * 8), d_hi = (unsigned)((b4 - b_seam) * 8);
      const unsigned    o_seam = gp.off8[G4SEAM] + (gp.seam_hi ? d_hi : d_lo);
      unsigned          keep;
      asm volatile("s_mov_b32 %[keep], m0\n\t"
                   "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b1]\n\t"
                   "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b1]\n\t"
                   "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b2]\n\t"
                   "s_add_u32 m0, %[lds], 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b2]\n\t"
                   "s_add_u32 m0, %[lds], 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b3]\n\t"
                   "s_add_u32 m0, %[lds], 0x1400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[bs]\n\t"
                   "s_add_u32 m0, %[lds], 0x1800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v6], %[b4]\n\t"
                   "s_add_u32 m0, %[lds], 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v7], %[b4]\n\t"
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep)
                   : [lds] "s"(lds_base), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3), [b4] "s"(b4), [bs] "s"(b_seam),
                     [v0] "v"(gp.off8[0]), [v1] "v"(gp.off8[1]), [v2] "v"(gp.off8[2]), [v3] "v"(gp.off8[3]), [v4] "v"(gp.off8[4]),
                     [v5] "v"(o_seam), [v6] "v"(gp.off8[6]), [v7] "v"(gp.off8[7])
                   : "memory", "scc");
    }

    // plane 0 of cell layer cz straight from memory (first layer of a chunk): the lane's row, 5 columns
    __device__ __forceinline__ void
    load_top(const DevGrid &g, const CellLane &c, const double *__restrict__ src, const int cz, double (&top)[N1])
    {
      const long long off = plane_offset(g, P * cz) + c.rowoff;
#pragma unroll
      for (int i = 0; i < N1; ++i)
        top[i] = src[off + i + (i == P ? c.xw4 : 0)];
    }

    // layout B (lane = y row `sub`, v[k*5 + i]): plane 0 from `top`, planes 1..4 from the staged image;
    // afterwards `top` holds plane 4 (= plane 0 of the next layer).  MASK: constrained DoFs read as zero.
    template <bool MASK>
    __device__ __forceinline__ void
    staged_read4(const DevGrid &g, const CellLane &c, const double *Gw, const int cz, double (&v)[NE], double (&top)[N1])
    {
      const double *base = Gw + c.sub * GROW + P * c.cl;
#pragma unroll
      for (int i = 0; i < N1; ++i)
        v[i] = top[i];
#pragma unroll
      for (int k = 1; k < N1; ++k)
        {
          const double *row = base + (k - 1) * (N1 * GROW);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            v[k * 5 + i] = row[i];
        }
#pragma unroll
      for (int i = 0; i < N1; ++i)
        top[i] = v[P * 5 + i];
      if (MASK)
        {
          const unsigned zd   = (g.dirichlet(2, P * cz) ? 1u : 0u) | (g.dirichlet(2, P * cz + P) ? 16u : 0u);
          const bool     mine = (c.yd >> c.sub) & 1u;
          if (c.cell_xy && (mine || zd != 0u || c.xd != 0u))
            {
#pragma unroll
              for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int i = 0; i < N1; ++i)
                  if (mine || (((c.xd >> i) | (zd >> s)) & 1u))
                    v[s * 5 + i] = 0.;
            }
        }
    }

    // 5-lane transpose: lane `sub` of a cell holds v[s*5 + f]; afterwards it holds
    // v'[s*5 + f] = (value v[sub*5 + f] of the cell's lane s), i.e. the lane index and the
    // slow register index change places (layout A <-> B).  One fast index f per round
    // through the wave-private buffer (320 doubles).
    // The LDS executes the instructions of one wave in order, so a read sees the partner lanes'
    // earlier writes and a later write cannot overtake an earlier read of the same words: no
    // s_waitcnt is needed between the rounds, only the compiler must not reorder the accesses.
    __device__ __forceinline__ void
    transpose5(double (&v)[NE], double *__restrict__ Tw, const CellLane &c)
    {
#if defined(ASM_ABLATE) && (ASM_ABLATE & 64)
      return; // diagnostic build (results are wrong)
#endif
#define TRANSPOSE_FENCE() asm volatile("" ::: "memory")
      // lanes 60..63 hold no cell: they run along unpredicated (their cl = 12 slots exist)
#pragma unroll
      for (int f = 0; f < N1; ++f)
        {
          double *buf = Tw;
#pragma unroll
          for (int s = 0; s < N1; ++s)
            buf[c.lane * N1 + s] = v[s * 5 + f];
          TRANSPOSE_FENCE();
#pragma unroll
          for (int s = 0; s < N1; ++s)
            v[s * 5 + f] = buf[(c.cl * N1 + s) * N1 + c.sub];
          TRANSPOSE_FENCE();
        }
#undef TRANSPOSE_FENCE
    }

    // value of lane - 5 (the same row / plane of the x-neighbour cell)
    __device__ __forceinline__ double
    lane_shift5(const double x, const int addr)
    {
      const unsigned long long u  = (unsigned long long)__double_as_longlong(x);
      const unsigned           lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(u & 0xffffffffull));
      const unsigned           hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(u >> 32));
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }

    // 16-byte accesses to 8-byte aligned rows
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef dbl2   dbl2_u __attribute__((aligned(8)));

    // ---------------------------------------------------------------------------
    // merge + store (layout B): o[k*5 + i] holds the lane's row of the cell's output tile.
    //   x: column 4 of the left cell is added to column 0 (5-lane shift inside the wave);
    //   z: plane 4 stays in the lane's registers (`carry`) and is added to plane 0 of the next layer;
    //   y: row 4 (lanes sub = 4) goes through a small LDS slot to the lanes sub = 0 of the wave
    //      above, one workgroup barrier per layer, slots double-buffered by layer parity.
    // After that every lane holds the final values of the DoFs it owns (columns 0..3 of its row,
    // planes 0..3) and applies the fused vector update while storing them: each DoF is
    // written once, no atomics, fixed summation order.
    // ---------------------------------------------------------------------------
    template <int MODE, int NPL>
    struct EpilogueOperands
    {
      static constexpr bool need_a = (MODE == OP_RESIDUAL || MODE == PC_STEP1 || MODE == PC_UPDATE);
      static constexpr bool need_b = (MODE == PC_UPDATE);
      double a[need_a ? NPL * P : 1];
      double b[need_b ? NPL * P : 1];
    };

    // operands of the fused update of planes K0 .. K0 + NPL - 1: issued early, used after the merge
    template <int MODE, int NPL>
    __device__ __forceinline__ void
    epilogue_prefetch(const LayerPlanes &lp, const CellLane &c, const TiledArgs &a, EpilogueOperands<MODE, NPL> &ops)
    {
      if (!EpilogueOperands<MODE, NPL>::need_a)
        return;
      const double *pa = (MODE == OP_RESIDUAL) ? a.b : a.x;
#if defined(ASM_ABLATE) && (ASM_ABLATE & 2)
      for (int e = 0; e < NPL * P; ++e)
        ops.a[e] = 1.0 + e;
      return;
#endif
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        {
          const long long off = lp.z[k] + c.rowoff;
#pragma unroll
          for (int i = 0; i < P; ++i)
            ops.a[k * P + i] = pa[off + i];
        }
    }
    // second operand of the three-term update (x_old): requested later than the first one, after the
    // sweeps (fewer registers are live there); the y-merge barrier covers part of its latency
    template <int MODE, int NPL>
    __device__ __forceinline__ void
    epilogue_prefetch_second(const LayerPlanes &lp, const CellLane &c, const TiledArgs &a, EpilogueOperands<MODE, NPL> &ops)
    {
      if (!EpilogueOperands<MODE, NPL>::need_b)
        return;
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        {
          const long long off = lp.z[k] + c.rowoff;
#pragma unroll
          for (int i = 0; i < P; ++i)
            ops.b[k * P + i] = a.x_old[off + i];
        }
    }

    template <int MODE>
    __device__ __forceinline__ double
    fused_update(const TiledArgs &a, const double t, const double oa, const double ob)
    {
      if (MODE == OP_PLAIN || MODE == PC_PLAIN)
        return t;
      if (MODE == OP_RESIDUAL)
        return oa - t;
      if (MODE == PC_FIRST)
        return a.f2 * t;
      if (MODE == PC_STEP1)
        return a.f1 * oa + a.f2 * t;
      return oa + a.f1 * (oa - ob) + a.f2 * t;
    }

    // x- and z-merge; on return o holds the x/z-complete rows, carry the top plane
    __device__ __forceinline__ void
    merge_xz(const CellLane &c, double (&o)[NE], double (&carry)[N1])
    {
      double r[N1];
#pragma unroll
      for (int k = 0; k < N1; ++k)
#if defined(ASM_ABLATE) && (ASM_ABLATE & 16)
        r[k] = o[k * 5 + N1 - 1];
#else
        r[k] = lane_shift5(o[k * 5 + N1 - 1], c.shl5);
#endif
      if (c.left_ok)
        {
#pragma unroll
          for (int k = 0; k < N1; ++k)
            o[k * 5] += r[k];
        }
#pragma unroll
      for (int i = 0; i < N1; ++i)
        {
          o[i] += carry[i];
          carry[i] = o[(N1 - 1) * 5 + i];
        }
    }

    template <int MODE, int NPL>
    __device__ __forceinline__ void
    merge_y_and_store(const DevGrid &g, const LayerPlanes &lp, const CellLane &c, const TiledArgs &a, double *__restrict__ Sy,
                      double (&o)[NE], const int K0, const EpilogueOperands<MODE, NPL> &ops STAMP_PARAM)
    {
      constexpr bool na = EpilogueOperands<MODE, NPL>::need_a, nb = EpilogueOperands<MODE, NPL>::need_b;
      // fused update + store of the lane's four owned values per plane
      const auto store_rows = [&]() {
#pragma unroll
        for (int k = 0; k < NPL; ++k)
          {
            const long long off = lp.z[k] + c.rowoff;
            double          rr[P];
#pragma unroll
            for (int i = 0; i < P; ++i)
              rr[i] = fused_update<MODE>(a, o[k * 5 + i], ops.a[na ? k * P + i : 0], ops.b[nb ? k * P + i : 0]);
#if defined(ASM_ABLATE) && (ASM_ABLATE & 4)
            if (rr[0] != 1.2345e300)
              continue;
#endif
            dbl2 lo, hi;
            lo.x = rr[0];
            lo.y = rr[1];
            hi.x = rr[2];
            hi.y = rr[3];
            // written once, not read again by this sweep: streaming stores for the FDM sweeps (round 2: -1.6 % per step on the
            // Dirichlet box, -4 % on the periodic one; round 4 re-measured: write-back stores cost them 1.3-2 %).  The Laplace
            // sweeps store through the L2 since round 4: -4.4 % at 80^3, -2.8 % at 128x128x96, -1.6 % on the periodic box
            // (their result is the residual the FDM sweep reads next, and a tile row of 11 cells = 352 B leaves half a 64-byte
            // sector to the neighbouring tile, which streaming stores write separately)
            if (MODE == OP_PLAIN || MODE == OP_RESIDUAL)
              {
                *(dbl2_u *)(a.dst + off)     = lo;
                *(dbl2_u *)(a.dst + off + 2) = hi;
              }
            else
              {
                __builtin_nontemporal_store(lo, (dbl2_u *)(a.dst + off));
                __builtin_nontemporal_store(hi, (dbl2_u *)(a.dst + off + 2));
              }
          }
      };
      // Only the rows sub = 0 receive something in the y merge: every other row is final before the barrier and is
      // stored there, where the wave would otherwise only wait - the eight waves of the workgroup then do not all enter
      // the vector-memory path together right after the barrier (profiles/r02_tiled_phase_stamps.txt).
      // Measured: the Laplace sweeps gain 6-9 %, the FDM sweeps (whose second update operand is requested late and
      // would have to be waited for before the barrier) lose 2-13 % in the modes with update operands.
      // (the FDM modes without update operands gain as well: -1 % Dirichlet, -4.5 % periodic box)
      constexpr bool split = (MODE == OP_PLAIN || MODE == OP_RESIDUAL || MODE == PC_PLAIN || MODE == PC_FIRST);
      const bool     late  = !split || (c.below_ok && c.sub == 0);
#if !(defined(ASM_ABLATE) && (ASM_ABLATE & 8))
      // ---- y: row 4 -> row 0 of the cell above
      if (c.cell_xy && c.sub == N1 - 1)
        {
          double *slot = Sy + (c.w * WX + c.cl) * YS;
#pragma unroll
          for (int k = 0; k < NPL; ++k)
#pragma unroll
            for (int i = 0; i < N1; ++i)
              slot[k * 5 + i] = o[k * 5 + i];
        }
      STAMP(10);
      if (split)
        {
          // the update operands (and the next layer's staged rows, requested long ago) have arrived
          if (na)
            gather_wait();
          if (c.own_row && !late)
            store_rows();
        }
      __syncthreads();
      STAMP(11);
      if (c.below_ok && c.sub == 0)
        {
          const double *slot = Sy + ((c.w - 1) * WX + c.cl) * YS;
#pragma unroll
          for (int k = 0; k < NPL; ++k)
#pragma unroll
            for (int i = 0; i < N1; ++i)
              o[k * 5 + i] += slot[k * 5 + i];
        }
#endif
      STAMP(12);
      // the next layer's staged rows (requested long ago) and the update operands have arrived;
      // waiting here, before this layer's stores are issued, keeps the stores out of the wait
      gather_wait();
      STAMP(13);
      // ---- fused update + store
      if (c.own_row)
        {
          const bool yd = (c.yd >> c.sub) & 1u;
          // all stores are issued back to back; constrained DoFs (boundary cells only) are put
          // right afterwards: their source values would otherwise make every plane's stores wait
          // for the plane before
          if (late)
            store_rows();
          // constrained DoFs: zero (plain product) or the identity row (fused modes:
          // (A x)_c = x_c, matrix_free_internal.h:247-255; unit rows of P^-1)
          const unsigned zdm = (g.dirichlet(2, K0) ? 1u : 0u) | ((NPL == P && g.dirichlet(2, K0 + P)) ? 16u : 0u);
          if (c.own_c4)
            {
              // the block's last column (dst may alias x_old: every operand is read before the store)
#pragma unroll
              for (int k = 0; k < NPL; ++k)
                {
                  const long long off = lp.z[k] + c.rowoff + P;
                  const bool      con = yd || ((zdm >> k) & 1u) || ((c.xd >> P) & 1u);
                  double          t   = o[k * 5 + P];
                  if (con)
                    t = (MODE == OP_PLAIN || MODE == PC_PLAIN) ? 0. : a.src[off];
                  const double oa = na ? ((MODE == OP_RESIDUAL) ? a.b : a.x)[off] : 0.;
                  const double ob = nb ? a.x_old[off] : 0.;
                  a.dst[off]      = fused_update<MODE>(a, t, oa, ob);
                }
            }
          if (yd || (c.xd & 0xfu) != 0u || (zdm & ((1u << NPL) - 1u)) != 0u)
            {
#pragma unroll
              for (int k = 0; k < NPL; ++k)
                {
                  const unsigned cm = (yd || ((zdm >> k) & 1u)) ? 0xfu : (c.xd & 0xfu);
                  if (cm != 0u)
                    {
                      const long long off = lp.z[k] + c.rowoff;
#pragma unroll
                      for (int i = 0; i < P; ++i)
                        if ((cm >> i) & 1u)
                          {
                            const double t = (MODE == OP_PLAIN || MODE == PC_PLAIN) ? 0. : a.src[off + i];
                            a.dst[off + i] = fused_update<MODE>(a, t, ops.a[na ? k * P + i : 0], ops.b[nb ? k * P + i : 0]);
                          }
                    }
                }
            }
        }
    }

    // ---------------------------------------------------------------------------
    // FDM sweep
    // ---------------------------------------------------------------------------
    template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
    __global__ void __launch_bounds__(NT, TY >= 8 ? TY / 4 : 2)
    fdm_tiled_kernel(const DevGrid g, const DevFdm f, const TiledArgs a, const Tiling tl)
    {
      WGTRACE_BEGIN(1)
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_T   = lds;
      double *S_Y   = S_T + LDS_T;
      double *S_G   = S_Y + LDS_Y;
      double *S_tab = S_G + LDS_G;

      int tx, ty, zc;
      workgroup_of_block(a, tl, tx, ty, zc);
      const int X0 = tx * OX + tl.x_first, Y0 = ty * OY + tl.y_first; // tile cell (0,0)
      // The waves above the block's last cell row (tiles of the last tile row: 80 rows = 8 + 10 x 7 + 2) have no cell: they
      // end here.  s_barrier waits for the surviving waves of a workgroup only, so the others go on as a smaller workgroup
      // (fewer waves at the LDS and in the vector-memory path per layer); the cooperative table loads below run over the
      // surviving threads.
      // INVARIANT for every edit below this line (all three tiled kernels): a block-wide cooperative loop strides by
      // nt_active, never by NT / blockDim.x, and no LDS hand-off may expect data from a wave at or above nt_active / 64 -
      // those waves are gone.  The diagnostic builds (-DASM_STAMP, workgroup trace) follow the same rule;
      // -DASM_EARLY_EXIT=0 keeps all waves alive (the A/B build of profiles/r03_tiled_kernel_experiments.txt).
#ifndef ASM_EARLY_EXIT
#  define ASM_EARLY_EXIT 1
#endif
      const int nt_active = ASM_EARLY_EXIT ? min(TY, g.nc[1] - Y0) * 64 : NT;
      if ((int)threadIdx.x >= nt_active)
        return;

      CellLane c;
      cell_lane_setup(g, X0, Y0, c);
      double *Tw = S_T + c.w * TBUF;
      double *Gw = S_G + c.w * GBUF;
      GatherPlan gp;
      gather_plan_setup(g, X0, Y0, gp);

      double *S_inv = S_tab + LDS_TAB;
      // eigen tables of tensor-product meshes live in LDS (the launcher checks n_inst <= 16)
      if (TENSOR)
        for (int e = threadIdx.x; e < f.n_inst * EIG; e += nt_active)
          S_tab[e] = f.eig[e];
      __syncthreads();

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;

      double v[NE]; // the lane's rows (layout B)
      double carry[N1] = {0., 0., 0., 0., 0.};
      int    par       = 0;
      gather_issue(g, gp, a.src, z_first, Gw);
      gather_wait();
      STAMP_DECL(16)
      const int cxs = c.cell_xy ? c.cx : 0, cys = c.cell_xy ? c.cy : 0; // table rows that exist
      // tensor-product meshes: the x and y instances of a lane do not change along z
      const bool active = c.cell_xy;
      int        ix_t = 0, iy_t = 0;
      if (TENSOR && active)
        {
          ix_t = f.inst[0][c.cx + 1];
          iy_t = f.inst[1][c.cy + 1];
        }
      const int iy_u = TENSOR ? uniform_load(f.inst[1], min(max(Y0 + (int)(threadIdx.x >> 6), -1), g.nc[1]) + 1) : 0;
      // the x and y factors of the weights are kept in registers: loading them layer by layer
      // would put vector loads (and waits for everything issued before them) into the layer
      double wpre_x[N1], wpre_y = 1., wpost_x[N1], wpost_y = 1.;
      if (HAS_PRE)
        {
          wpre_y = f.w_pre[1][(cys + 1) * N1 + c.sub];
#pragma unroll
          for (int i = 0; i < N1; ++i)
            wpre_x[i] = f.w_pre[0][(cxs + 1) * N1 + i];
        }
      if (HAS_POST)
        {
          wpost_y = f.w_post[1][(cys + 1) * N1 + c.sub];
#pragma unroll
          for (int i = 0; i < N1; ++i)
            wpost_x[i] = f.w_post[0][(cxs + 1) * N1 + i];
        }
      const bool inv_lds   = TENSOR && f.inv_table != nullptr && f.n_inst <= INV_MAX;
      int        iz_cached = -1;
      for (int cz = z_first; cz < Ze; ++cz)
        {
          STAMP(0);
          const LayerPlanes lp = layer_planes(g, cz);
          staged_read<true, false>(g, c, Gw, cz, v);
          gather_release();
          int ix = ix_t, iy = iy_t, iz = 0;
          if (TENSOR)
            iz = active ? uniform_load(f.inst[2], cz + 1) : 0;
          else if (active)
            {
              const long long ce =
                (c.cx + 1) + (long long)(g.nc[0] + 2) * ((c.cy + 1) + (long long)(g.nc[1] + 2) * (cz + 1));
              ix = f.inst[0][ce];
              iy = f.inst[1][ce];
              iz = f.inst[2][ce];
            }
          // two address spaces, two pointer sets: a merged pointer would compile to flat loads
          const double *exl = S_tab + ix * EIG, *eyl = S_tab + iy * EIG, *ezl = S_tab + iz * EIG;
          const double *exg = f.eig + (long long)ix * EIG, *eyg = f.eig + (long long)iy * EIG,
                       *ezg = f.eig + (long long)iz * EIG;
          // tensor-product meshes: the y and z matrices depend on the cell row / layer only
          // (taken from the row / layer index, not from a lane: lane 0 may hold a cell that does not exist)
          const int iz_u = TENSOR ? uniform_load(f.inst[2], cz + 1) : 0;
          // reciprocal eigenvalue sums of all (ix, iy) pairs of this layer's z instance: kept in LDS,
          // refilled when the z instance changes (first / last layers of a block)
          if (inv_lds && iz_u != iz_cached)
            {
              __syncthreads(); // the previous tables are no longer read
              const int nn = f.n_inst * f.n_inst * (N1 * NE);
              for (int e = threadIdx.x; e < nn; e += nt_active)
                S_inv[e] = f.inv_table[(long long)iz_u * nn + e];
              __syncthreads();
              iz_cached = iz_u;
            }
          const UniformMatrix Uy(f.eig + (long long)__builtin_amdgcn_readfirstlane(iy_u) * EIG),
            Uz(f.eig + (long long)__builtin_amdgcn_readfirstlane(iz_u) * EIG);
          // the x matrix is wave-uniform too unless the cell row touches a boundary column
          unsigned long long amask = __ballot(active);
          const int          lane0 = amask ? __ffsll((long long)amask) - 1 : 0;
          const int          ix_u  = __builtin_amdgcn_readlane(ix, lane0);
          const bool         x_uniform = TENSOR && __all(!active || ix == ix_u);
          const UniformMatrix Ux(f.eig + (long long)ix_u * EIG);
          // parity-sorted eigenvectors (cells without a Dirichlet face): even-odd sweeps
          const bool sym_x = x_uniform && uniform_load(f.eig_sym, ix_u) != 0;
          const bool sym_y = TENSOR && uniform_load(f.eig_sym, iy_u) != 0;
          const bool sym_z = TENSOR && uniform_load(f.eig_sym, iz_u) != 0;
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
          // offset 0); nothing they compute is ever merged or stored.
          // ---- layout B: lane = (cell, y row j), v[k*5 + i]; x and z forward sweeps -------
          if (HAS_PRE)
            {
#pragma unroll
              for (int k = 0; k < N1; ++k)
                {
                  const double wyz = uniform_load(f.w_pre[2], (cz + 1) * N1 + k) * wpre_y;
#pragma unroll
                  for (int i = 0; i < N1; ++i)
                    v[k * 5 + i] *= wpre_x[i] * wyz;
                }
            }
          if (sym_x)
            sweep_fast_eo<true>(v, Ux.fresh());
          else
            {
              LOADQ(q, ex);
              sweep_fast<true>(v, q);
            }
          if (sym_z)
            sweep_slow_eo<true>(v, Uz.fresh());
          else if (TENSOR)
            sweep_slow<true>(v, Uz.fresh());
          else
            {
              LOADQ(q, ez);
              sweep_slow<true>(v, q);
            }
          // ---- layout A: lane = (cell, z index c), v[b*5 + a]; y sweeps + scaling ---------
          STAMP(2);
          transpose5(v, Tw, c);
          STAMP(3);
          if (sym_y)
            sweep_slow_eo<true>(v, Uy.fresh());
          else if (TENSOR)
            sweep_slow<true>(v, Uy.fresh());
          else
            {
              LOADQ(q, ey);
              sweep_slow<true>(v, q);
            }
#if defined(ASM_ABLATE) && (ASM_ABLATE & 128)
          if (false)
            {
            }
#else
          if (inv_lds)
            {
              const double *inv = S_inv + (iy * f.n_inst + ix) * (N1 * NE) + c.sub * NE;
#pragma unroll
              for (int e = 0; e < NE; ++e)
                v[e] *= inv[e];
            }
#endif
          else if (f.inv_table != nullptr)
            {
              // table [iz][iy][ix][c][b][a]
              const double *inv = f.inv_table + (long long)((iz * f.n_inst + iy) * f.n_inst + ix) * (N1 * NE) + c.sub * NE;
#pragma unroll
              for (int e = 0; e < NE; ++e)
                v[e] *= inv[e];
            }
#if defined(ASM_ABLATE) && (ASM_ABLATE & 128)
          else if (false)
#else
          else
#endif
            {
              const double lzc = TENSOR ? ezl[NE + c.sub] : ezg[NE + c.sub];
#pragma unroll
              for (int bb = 0; bb < N1; ++bb)
#pragma unroll
                for (int aa = 0; aa < N1; ++aa)
                  {
                    // reciprocal with two Newton steps (full double precision)
                    const double sm = (TENSOR ? exl[NE + aa] : exg[NE + aa]) + ((TENSOR ? eyl[NE + bb] : eyg[NE + bb]) + lzc);
                    double       r  = __builtin_amdgcn_rcp(sm);
                    r               = fma(fma(-sm, r, 1.0), r, r);
                    r               = fma(fma(-sm, r, 1.0), r, r);
                    v[bb * 5 + aa] *= r;
                  }
            }
          // request the next layer's rows and this layer's update operands: behind the last vector
          // loads of the layer (eigenvalue table), so that nothing has to wait for them before the
          // merge; they have until the store phase / the next layer to arrive
#if !(defined(ASM_ABLATE) && (ASM_ABLATE & 1))
          if (cz + 1 < Ze)
            gather_issue(g, gp, a.src, cz + 1, Gw);
#endif
          EpilogueOperands<MODE, P> ops;
          if (cz >= Zb)
            epilogue_prefetch<MODE, P>(lp, c, a, ops);
          if (sym_y)
            sweep_slow_eo<false>(v, Uy.fresh());
          else if (TENSOR)
            sweep_slow<false>(v, Uy.fresh());
          else
            sweep_slow<false>(v, q);
          // ---- back to layout B; z and x backward sweeps, post weights --------------------
          STAMP(4);
          transpose5(v, Tw, c);
          STAMP(5);
          if (sym_z)
            sweep_slow_eo<false>(v, Uz.fresh());
          else if (TENSOR)
            sweep_slow<false>(v, Uz.fresh());
          else
            {
              LOADQ(q, ez);
              sweep_slow<false>(v, q);
            }
          if (sym_x)
            sweep_fast_eo<false>(v, Ux.fresh());
          else
            {
              LOADQ(q, ex);
              sweep_fast<false>(v, q);
            }
          if (HAS_POST)
            {
#pragma unroll
              for (int k = 0; k < N1; ++k)
                {
                  const double wyz = uniform_load(f.w_post[2], (cz + 1) * N1 + k) * wpost_y;
#pragma unroll
                  for (int i = 0; i < N1; ++i)
                    v[k * 5 + i] *= wpost_x[i] * wyz;
                }
            }
          STAMP(6);
          merge_xz(c, v, carry);
          STAMP(7);
          if (cz >= Zb)
            {
              epilogue_prefetch_second<MODE, P>(lp, c, a, ops);
              merge_y_and_store<MODE, P>(g, lp, c, a, S_Y + par * (LDS_Y / 2), v, P * cz, ops STAMP_ARG);
              par ^= 1;
            }
          else
            gather_wait(); // halo layer: nothing stored, only the carry leaves it
          STAMP(8);
        }
      STAMP_FLUSH();
#undef LOADQ
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        {
          // top plane of the block (physical boundary): what the last layer left in the carry
          EpilogueOperands<MODE, 1> ops;
          const LayerPlanes         lp = layer_planes(g, g.nc[2]);
          epilogue_prefetch<MODE, 1>(lp, c, a, ops);
          epilogue_prefetch_second<MODE, 1>(lp, c, a, ops);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            v[i] = carry[i];
          merge_y_and_store<MODE, 1>(g, lp, c, a, S_Y + par * (LDS_Y / 2), v, P * g.nc[2], ops STAMP_ARG);
        }
      WGTRACE_END();
    }

    // ---------------------------------------------------------------------------
    // Laplace sweep on Cartesian cells.  For a box cell the sum-factorised evaluate / integrate
    // pair with exact Gauss quadrature (include/operator.h:866-875) equals the Kronecker sum
    //   hy hz/hx Mz My Kx + hx hz/hy Mz Ky Mx + hx hy/hz Kz My Mx
    // of the 1-D reference mass / stiffness matrices.  With their simultaneous diagonalisation
    // M = F F^T, K = F diag(lam) F^T (numerics.h: Basis1D::diagonalise) the cell matrix is
    //   (F x F x F) diag(hy hz/hx lam_a + hx hz/hy lam_b + hx hy/hz lam_c) (F x F x F)^T :
    // 6 one-dimensional sweeps with ONE wave-uniform matrix whose columns are even / odd vectors
    // (even-odd form: 17 operations per line), a diagonal scaling in between, no quadrature-point
    // data, and the same register choreography as the FDM sweep.
    // ---------------------------------------------------------------------------
    template <int MODE>
    __global__ void __launch_bounds__(NT, TY >= 8 ? TY / 4 : 2)
    laplace_cart_tiled_kernel(const DevGrid g, const DevOp op, const TiledArgs a, const Tiling tl)
    {
      WGTRACE_BEGIN(0)
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_T = lds;
      double *S_Y = S_T + LDS_T;
      double *S_G = S_Y + LDS_Y;

      int tx, ty, zc;
      workgroup_of_block(a, tl, tx, ty, zc);
      const int X0 = tx * OX + tl.x_first, Y0 = ty * OY + tl.y_first;
      // (waves above the block's last cell row end here: see fdm_tiled_kernel)
      if (ASM_EARLY_EXIT && (int)threadIdx.x >= min(TY, g.nc[1] - Y0) * 64)
        return;

      CellLane c;
      cell_lane_setup(g, X0, Y0, c);
      double *Tw = S_T + c.w * TBUF;
      double *Gw = S_G + c.w * GBUF;
      GatherPlan gp;
      gather_plan_setup(g, X0, Y0, gp);

      // F (row i, column a) and the eigenvalues: wave-uniform scalar operands
      const UniformMatrix Fm(op.F_ref);
      const UniformMatrix Lm(op.lam_ref);

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;
      double    hx = 1., hy = 1.;
      if (c.cell_xy)
        {
          hx = op.h[0][c.cx + 1];
          hy = op.h[1][c.cy + 1];
        }
      const double lam_sub = op.lam_ref[c.sub]; // eigenvalue of the lane's own index in the middle layout
      const double a_xy = hy / hx, b_xy = hx / hy, g_xy = hx * hy; // layer-invariant parts of the three cell factors

      double u[NE]; // the lane's rows (layout B)
      double carry[N1] = {0., 0., 0., 0., 0.};
      int    par       = 0;
      gather_issue(g, gp, a.src, z_first, Gw);
      gather_wait();
      STAMP_DECL(0)
      for (int cz = z_first; cz < Ze; ++cz)
        {
          STAMP(0);
          const LayerPlanes lp = layer_planes(g, cz);
          staged_read<true, true>(g, c, Gw, cz, u);
          gather_release();
          const double hz    = uniform_load(op.h[2], cz + 1);
          const double alpha = a_xy * hz, beta = b_xy * hz, gamma = g_xy / hz;
          // lanes without a cell run along unpredicated; nothing of theirs is merged or stored
          // ---- layout B: lane = (cell, y row j), u[k*5 + i]; x and z forward sweeps -------
          STAMP(1);
          sweep_fast_eo<true>(u, Fm);
          sweep_slow_eo<true>(u, Fm);
          // the staged image has been consumed: request the next layer's rows
          const GatherRequest rq(g, gp, a.src, cz + 1, Gw, cz + 1 < Ze);
          rq.issue();
          // ---- layout A: lane = (cell, z index c), u[b*5 + a]; y sweeps + scaling ---------
          STAMP(2);
          transpose5(u, Tw, c);
          STAMP(3);
          sweep_slow_eo<true>(u, Fm);
          // this layer's update operands (halo layers load them too: no branch around the loads)
          EpilogueOperands<MODE, P> ops;
          epilogue_prefetch<MODE, P>(lp, c, a, ops);
          {
            const double tc = gamma * lam_sub;
#pragma unroll
            for (int bb = 0; bb < N1; ++bb)
              {
                const double tbc = fma(beta, Lm[bb], tc);
#pragma unroll
                for (int aa = 0; aa < N1; ++aa)
                  u[bb * 5 + aa] *= fma(alpha, Lm[aa], tbc);
              }
          }
          sweep_slow_eo<false>(u, Fm);
          // ---- back to layout B; z and x backward sweeps ------------------------------------
          STAMP(4);
          transpose5(u, Tw, c);
          STAMP(5);
          sweep_slow_eo<false>(u, Fm);
          sweep_fast_eo<false>(u, Fm);
          STAMP(6);
          merge_xz(c, u, carry);
          STAMP(7);
          if (cz >= Zb)
            {
              merge_y_and_store<MODE, P>(g, lp, c, a, S_Y + par * (LDS_Y / 2), u, P * cz, ops STAMP_ARG);
              par ^= 1;
            }
          else
            gather_wait(); // halo layer: nothing stored, only the carry leaves it
          STAMP(8);
        }
      STAMP_FLUSH();
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        {
          EpilogueOperands<MODE, 1> ops;
          const LayerPlanes         lp = layer_planes(g, g.nc[2]);
          epilogue_prefetch<MODE, 1>(lp, c, a, ops);
          epilogue_prefetch_second<MODE, 1>(lp, c, a, ops);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            u[i] = carry[i];
          merge_y_and_store<MODE, 1>(g, lp, c, a, S_Y + par * (LDS_Y / 2), u, P * g.nc[2], ops STAMP_ARG);
        }
      WGTRACE_END();
    }


    // ---------------------------------------------------------------------------
    // Second-generation Laplace sweep: the same cell algebra and merge choreography as
    // laplace_cart_tiled_kernel, with TYW cell rows (waves) per workgroup instead of 8, the source
    // plane shared by two consecutive layers kept in registers (staged_read4) and a register budget of
    // three waves per SIMD: more rows per tile mean fewer halo rows and a better fit of the tiles to
    // the mesh, the third wave covers the latencies two waves leave exposed.
    // ---------------------------------------------------------------------------
    template <int TYW>
    struct Tile2
    {
      static constexpr int NTW   = TYW * 64;
      static constexpr int OYW   = TYW - 1;
      static constexpr int L_T   = TYW * TBUF;
      static constexpr int L_Y   = 2 * TYW * YBUF;
      static constexpr int L_G   = TYW * G4BUF;
      static constexpr int TOTAL = L_T + L_Y + L_G;
      static_assert(TOTAL * sizeof(double) <= 160 * 1024, "one workgroup must fit the 160 KB LDS of a CU");
    };

    template <int MODE, int TYW>
    __global__ void __launch_bounds__(TYW * 64)
    laplace_cart_tiled_kernel2(const DevGrid g, const DevOp op, const TiledArgs a, const Tiling tl)
    {
      using T2 = Tile2<TYW>;
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_T = lds;
      double *S_Y = S_T + T2::L_T;
      double *S_G = S_Y + T2::L_Y;

      int tx, ty, zc;
      workgroup_of_block(a, tl, tx, ty, zc);
      const int X0 = tx * OX + tl.x_first, Y0 = ty * T2::OYW + tl.y_first;
      // (waves above the block's last cell row end here: see fdm_tiled_kernel)
      if (ASM_EARLY_EXIT && (int)threadIdx.x >= min(TYW, g.nc[1] - Y0) * 64)
        return;

      CellLane c;
      cell_lane_setup(g, X0, Y0, c);
      double *Tw = S_T + c.w * TBUF;
      double *Gw = S_G + c.w * G4BUF;
      GatherPlan4 gp;
      gather4_plan_setup(g, X0, Y0, gp);

      const UniformMatrix Fm(op.F_ref);
      const UniformMatrix Lm(op.lam_ref);

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;
      double    hx = 1., hy = 1.;
      if (c.cell_xy)
        {
          hx = op.h[0][c.cx + 1];
          hy = op.h[1][c.cy + 1];
        }
      const double lam_sub = op.lam_ref[c.sub];
      const double a_xy = hy / hx, b_xy = hx / hy, g_xy = hx * hy;

      double u[NE];
      double top[N1];
      double carry[N1] = {0., 0., 0., 0., 0.};
      int    par       = 0;
      STAMP_DECL(0) // (diagnostic builds: shares the first-generation kernel's slots)
      gather4_issue(g, gp, a.src, z_first, Gw, true);
      load_top(g, c, a.src, z_first, top);
      gather_wait();
      for (int cz = z_first; cz < Ze; ++cz)
        {
          const LayerPlanes lp = layer_planes(g, cz);
          staged_read4<true>(g, c, Gw, cz, u, top);
          gather_release();
          const double hz    = uniform_load(op.h[2], cz + 1);
          const double alpha = a_xy * hz, beta = b_xy * hz, gamma = g_xy / hz;
          sweep_fast_eo<true>(u, Fm);
          sweep_slow_eo<true>(u, Fm);
          gather4_issue(g, gp, a.src, cz + 1, Gw, cz + 1 < Ze);
          transpose5(u, Tw, c);
          sweep_slow_eo<true>(u, Fm);
          // this layer's update operands (halo layers load them too: no branch around the loads)
          EpilogueOperands<MODE, P> ops;
          epilogue_prefetch<MODE, P>(lp, c, a, ops);
          {
            const double tc = gamma * lam_sub;
#pragma unroll
            for (int bb = 0; bb < N1; ++bb)
              {
                const double tbc = fma(beta, Lm[bb], tc);
#pragma unroll
                for (int aa = 0; aa < N1; ++aa)
                  u[bb * 5 + aa] *= fma(alpha, Lm[aa], tbc);
              }
          }
          sweep_slow_eo<false>(u, Fm);
          transpose5(u, Tw, c);
          sweep_slow_eo<false>(u, Fm);
          sweep_fast_eo<false>(u, Fm);
          merge_xz(c, u, carry);
          if (cz >= Zb)
            {
              merge_y_and_store<MODE, P>(g, lp, c, a, S_Y + par * (T2::L_Y / 2), u, P * cz, ops STAMP_ARG);
              par ^= 1;
            }
          else
            gather_wait();
        }
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        {
          EpilogueOperands<MODE, 1> ops;
          const LayerPlanes         lp = layer_planes(g, g.nc[2]);
          epilogue_prefetch<MODE, 1>(lp, c, a, ops);
          epilogue_prefetch_second<MODE, 1>(lp, c, a, ops);
#pragma unroll
          for (int i = 0; i < N1; ++i)
            u[i] = carry[i];
          merge_y_and_store<MODE, 1>(g, lp, c, a, S_Y + par * (T2::L_Y / 2), u, P * g.nc[2], ops STAMP_ARG);
        }
    }


    Tiling
    make_tiling(const DevGrid &g, const TiledArgs &a, const int oy = OY)
    {
      int    n_cu = a.n_cu;
      Tiling tl;
      tl.x_first = (g.wrap[0] || g.ext_lo[0] < 0) ? -1 : 0; // (cell_computable(g, d, -1))
      tl.y_first = (g.wrap[1] || g.ext_lo[1] < 0) ? -1 : 0;
      // tile t covers cells t OX + x_first ... + OX: the last one must hold cell nc - 1
      tl.ntx = std::max(1, (g.nc[0] - 1 - tl.x_first + OX - 1) / OX);
      tl.nty = std::max(1, (g.nc[1] - 1 - tl.y_first + oy - 1) / oy);
      tl.n_box = tl.n_first = 0;
      // the last owned plane nc*p of a non-wrapped direction is produced by the tile
      // that holds cell nc-1, no extra tile needed
      const int tiles = tl.ntx * tl.nty;
      // One workgroup per CU at a time; a chunk of zlen layers costs zlen + 1 layer times (halo layer) plus about half a
      // layer of start-up.  The workgroups are handed out as compute units fall free and live up to 1.35 times as long
      // at Dirichlet faces as inside (profiles/r03_workgroup_trace.txt), so the launch takes between work / n_cu and
      // whole rounds of the longest workgroup, plus a tail of about half a workgroup: the chunk count with the smallest
      //   (ceil(N / n_cu) + N / n_cu) / 2 x L + tail x L,   N = tiles x chunks, L = zlen + 1.5
      // Round 2's model, rounds x L alone, took ONE chunk of 96 layers for 128 x 128 x 96 cells (228 workgroups on 256
      // compute units in a single round, no balancing at all): P 0.98 ms where four to twelve chunks give 0.76-0.80
      // (profiles/r03_zchunks_north_star.txt); at 80^3 both models take 8 chunks.
      if (n_cu <= 0)
        n_cu = 256;
      // (a block without Dirichlet faces has workgroups of one length: no tail term; the periodic 128 x 64 x 64 box runs
      //  its 240 workgroups of two chunks in one round, 0.50 ms per step against 0.53 with eight chunks)
      static const double tail_env = std::getenv("ASM_TILED_TAIL") ? std::atof(std::getenv("ASM_TILED_TAIL")) : 0.5; // experiments
      bool                uneven   = false;
      for (int d = 0; d < 3; ++d)
        uneven = uneven || g.dir_lo[d] || g.dir_hi[d];
      const double tail = uneven ? tail_env : 0.;
      int    best      = 1;
      double best_cost = 1e300;
      for (int nzc = 1; nzc <= g.nc[2]; ++nzc)
        {
          const int    zlen   = (g.nc[2] + nzc - 1) / nzc;
          const int    chunks = (g.nc[2] + zlen - 1) / zlen;
          const double N      = (double)tiles * chunks, L = zlen + 1.5;
          const double cost   = 0.5 * (std::ceil(N / n_cu) + N / n_cu) * L + tail * L;
          if (cost < best_cost - 1e-9)
            {
              best_cost = cost;
              best      = nzc;
            }
        }
      // Two-phase launch (ghost exchange of the result overlapped with phase 2): phase 1 = the workgroups that produce
      // what neighbours ghost, padded with interior ones to whole rounds; the exchange (about x_layers layer times)
      // has phase 2 to hide in.  More, shorter chunks make phase 1 short when the z faces are interfaces.
      const auto phases = [&](const int chunks, int &n_first, int &n_int_total) {
        const int ix = tl.ntx - (a.iface[0][0] ? 1 : 0) - (a.iface[0][1] ? 1 : 0), iy = tl.nty - (a.iface[1][0] ? 1 : 0) - (a.iface[1][1] ? 1 : 0),
                  iz = chunks - (a.iface[2][0] ? 1 : 0) - (a.iface[2][1] ? 1 : 0);
        n_int_total  = std::max(ix, 0) * std::max(iy, 0) * std::max(iz, 0);
        const int nb = tiles * chunks - n_int_total;
        n_first      = std::min(tiles * chunks, (nb + n_cu - 1) / n_cu * n_cu);
      };
      if (a.phase != 0)
        {
          static const double x_layers = std::getenv("ASM_EXCHANGE_LAYERS") ? std::atof(std::getenv("ASM_EXCHANGE_LAYERS")) : 20.;
          best_cost                    = 1e300;
          for (int nzc = 1; nzc <= g.nc[2]; ++nzc)
            {
              const int zlen = (g.nc[2] + nzc - 1) / nzc, chunks = (g.nc[2] + zlen - 1) / zlen;
              int       n_first, n_int;
              phases(chunks, n_first, n_int);
              const double L    = zlen + 1.5;
              const double cost = (n_first + n_cu - 1) / n_cu * L + std::max((tiles * chunks - n_first + n_cu - 1) / n_cu * L, x_layers);
              if (cost < best_cost - 1e-9)
                {
                  best_cost = cost;
                  best      = nzc;
                }
            }
        }
      static const int forced = std::getenv("ASM_TILED_ZCHUNKS") ? std::atoi(std::getenv("ASM_TILED_ZCHUNKS")) : 0; // experiments only
      if (forced > 0)
        best = std::min(forced, g.nc[2]);
      tl.zlen = (g.nc[2] + best - 1) / best;
      tl.nzc  = (g.nc[2] + tl.zlen - 1) / tl.zlen;
      if (a.phase != 0)
        {
          int n_first, n_int;
          phases(tl.nzc, n_first, n_int);
          tl.n_first = n_first;
          const int n[3]  = {tl.ntx, tl.nty, tl.nzc};
          int       lo[3] = {0, 0, 0}, hi[3] = {n[0], n[1], n[2]}; // what is left of the box after the slabs taken so far
          const auto add = [&](const int blo[3], const int bhi[3]) {
            const int sz[3] = {bhi[0] - blo[0], bhi[1] - blo[1], bhi[2] - blo[2]};
            if (sz[0] <= 0 || sz[1] <= 0 || sz[2] <= 0)
              return;
            const int k = tl.n_box++;
            for (int d = 0; d < 3; ++d)
              {
                tl.box_lo[k][d]   = blo[d];
                tl.box_size[k][d] = sz[d];
              }
            tl.box_end[k] = (k ? tl.box_end[k - 1] : 0) + sz[0] * sz[1] * sz[2];
          };
          for (int d = 2; d >= 0; --d) // slabs: z low / high, then y, then x
            for (int f = 0; f < 2; ++f)
              if (a.iface[d][f] && hi[d] > lo[d])
                {
                  int blo[3] = {lo[0], lo[1], lo[2]}, bhi[3] = {hi[0], hi[1], hi[2]};
                  if (f == 0)
                    bhi[d] = ++lo[d];
                  else
                    blo[d] = --hi[d];
                  add(blo, bhi);
                }
          add(lo, hi); // the interior
        }
      return tl;
    }

    // blocks of a launch (0: nothing to do in this phase)
    int
    blocks_of_phase(const Tiling &tl, const TiledArgs &a)
    {
      const int total = tl.ntx * tl.nty * tl.nzc;
      return a.phase == 0 ? total : (a.phase == 1 ? tl.n_first : total - tl.n_first);
    }

    int
    tile_rows_of(const int rows)
    {
      return rows >= 9 && rows <= 11 ? rows - 1 : OY;
    }

    // The kernels need more dynamic LDS than the default limit: raise it once per kernel and device
    // (bit d of `done` = device d).  false: the attribute was refused, the launch would be rejected.
    template <typename K>
    bool
    set_lds_limit(K kernel, std::atomic<unsigned long long> &done, const int device, const size_t bytes = LDS_TOTAL * sizeof(double))
    {
      const unsigned long long bit = 1ull << (device & 63);
      if (done.load(std::memory_order_acquire) & bit)
        return true;
      if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        {
          (void)hipGetLastError();
          return false;
        }
      done.fetch_or(bit, std::memory_order_release);
      return true;
    }
  } // namespace

  void
  tiled_plan(const DevGrid &g, const int n_cu, const int rows, int out[8])
  {
    TiledArgs a{};
    a.phase         = 0;
    a.n_cu          = n_cu;
    const Tiling tl = make_tiling(g, a, tile_rows_of(rows));
    out[0] = tl.ntx;
    out[1] = tl.nty;
    out[2] = tl.nzc;
    out[3] = tl.zlen;
    out[4] = tl.x_first;
    out[5] = tl.y_first;
    out[6] = OX;
    out[7] = tile_rows_of(rows);
  }

  int
  tiled_order(const DevGrid &g, const int n_cu, const int rows, std::vector<int> &order)
  {
    order.clear();
    if (g.p != P)
      return 0;
    TiledArgs a{};
    a.phase      = 0;
    a.n_cu       = n_cu;
    const Tiling tl = make_tiling(g, a, tile_rows_of(rows));
    const int    n  = tl.ntx * tl.nty * tl.nzc;
    // extra life of a workgroup whose cells touch a Dirichlet face, in per cent of an interior one's (measured:
    // profiles/r03_workgroup_trace.txt; the low y face costs most: every DoF row of a lane is constrained there)
    static const int extra[3][2] = {{20, 13}, {33, 10}, {0, 6}};
    const auto cost = [&](const int i) {
      const int tx = i % tl.ntx, ty = (i / tl.ntx) % tl.nty, zc = i / (tl.ntx * tl.nty);
      const int t[3] = {tx, ty, zc}, nt[3] = {tl.ntx, tl.nty, tl.nzc};
      int       c    = 0;
      for (int d = 0; d < 3; ++d)
        c += (g.dir_lo[d] && t[d] == 0 ? extra[d][0] : 0) + (g.dir_hi[d] && t[d] == nt[d] - 1 ? extra[d][1] : 0);
      return c;
    };
    bool any = false;
    for (int i = 0; i < n && !any; ++i)
      any = cost(i) != 0;
    if (!any || n < 16)
      return 0;
    order.resize(n);
    // the contiguous range of every XCD (xcd_remap), expensive workgroups first, equally expensive ones in place
    for (int x = 0; x < 8; ++x)
      {
        const int begin = x * (n >> 3) + std::min(x, n & 7), size = (n >> 3) + (x < (n & 7) ? 1 : 0);
        for (int i = 0; i < size; ++i)
          order[begin + i] = begin + i;
        std::stable_sort(order.begin() + begin, order.begin() + begin + size, [&](const int u, const int v) { return cost(u) > cost(v); });
      }
    return n;
  }

  template <int MODE>
  static bool
  launch_laplace_tiled(const DevGrid &g, const DevOp &op, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    static std::atomic<unsigned long long> done{0};
    if (!set_lds_limit(laplace_cart_tiled_kernel<MODE>, done, a.device))
      return false;
    if (blocks_of_phase(tl, a) > 0)
      hipLaunchKernelGGL((laplace_cart_tiled_kernel<MODE>), dim3(blocks_of_phase(tl, a)), dim3(NT), LDS_TOTAL * sizeof(double),
                       st, g, op, a, tl);
    return true;
  }

  template <int MODE, int TYW>
  static bool
  launch_laplace_tiled2(const DevGrid &g, const DevOp &op, const TiledArgs &a, hipStream_t st)
  {
    using T2                                      = Tile2<TYW>;
    static std::atomic<unsigned long long> done{0};
    const size_t                           bytes = T2::TOTAL * sizeof(double);
    if (!set_lds_limit(laplace_cart_tiled_kernel2<MODE, TYW>, done, a.device, bytes))
      return false;
    const Tiling tl = make_tiling(g, a, T2::OYW);
    if (blocks_of_phase(tl, a) > 0)
      hipLaunchKernelGGL((laplace_cart_tiled_kernel2<MODE, TYW>), dim3(blocks_of_phase(tl, a)), dim3(T2::NTW), bytes, st, g, op, a, tl);
    return true;
  }

  bool
  laplace_tiled(const DevGrid &g, const DevOp &op, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || !op.cartesian)
      return false;
    // a.rows: cell rows (waves) per workgroup; 0 = first-generation kernel (8 rows).  Which one is the
    // fastest depends on how the tiles fit the mesh and the device: the context measures it once (context.hip)
    if (a.rows != 0 && (a.mode == OP_PLAIN || a.mode == OP_RESIDUAL))
      {
        const int rows = a.rows;
#define ASM_LAUNCH2(R)                                                  \
  case R:                                                               \
    return a.mode == OP_PLAIN ? launch_laplace_tiled2<OP_PLAIN, R>(g, op, a, st) : launch_laplace_tiled2<OP_RESIDUAL, R>(g, op, a, st);
        switch (rows)
          {
            ASM_LAUNCH2(9)
            ASM_LAUNCH2(10)
            ASM_LAUNCH2(11)
            default: return false;
          }
#undef ASM_LAUNCH2
      }
    const Tiling tl = make_tiling(g, a);
    if (a.mode == OP_PLAIN)
      return launch_laplace_tiled<OP_PLAIN>(g, op, a, tl, st);
    if (a.mode == OP_RESIDUAL)
      return launch_laplace_tiled<OP_RESIDUAL>(g, op, a, tl, st);
    return false;
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
  static bool
  launch_fdm_tiled_t(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    static std::atomic<unsigned long long> done{0};
    if (!set_lds_limit(fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>, done, a.device))
      return false;
    if (blocks_of_phase(tl, a) > 0)
      hipLaunchKernelGGL((fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>), dim3(blocks_of_phase(tl, a)), dim3(NT),
                       LDS_TOTAL * sizeof(double), st, g, f, a, tl);
    return true;
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST>
  static bool
  launch_fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    if (f.tensor_inst && f.n_inst <= 16)
      return launch_fdm_tiled_t<MODE, HAS_PRE, HAS_POST, true>(g, f, a, tl, st);
    return launch_fdm_tiled_t<MODE, HAS_PRE, HAS_POST, false>(g, f, a, tl, st);
  }

  template <bool HAS_PRE, bool HAS_POST>
  static bool
  dispatch_fdm_mode(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    switch (a.mode)
      {
        case PC_PLAIN: return launch_fdm_tiled<PC_PLAIN, HAS_PRE, HAS_POST>(g, f, a, tl, st);
        case PC_FIRST: return launch_fdm_tiled<PC_FIRST, HAS_PRE, HAS_POST>(g, f, a, tl, st);
        case PC_STEP1: return launch_fdm_tiled<PC_STEP1, HAS_PRE, HAS_POST>(g, f, a, tl, st);
        case PC_UPDATE: return launch_fdm_tiled<PC_UPDATE, HAS_PRE, HAS_POST>(g, f, a, tl, st);
        default: return false;
      }
  }


  bool
  fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || f.m != N1 || f.vertex)
      return false;
    const Tiling tl = make_tiling(g, a);
    if (f.has_pre && f.has_post)
      return dispatch_fdm_mode<true, true>(g, f, a, tl, st);
    if (f.has_pre)
      return dispatch_fdm_mode<true, false>(g, f, a, tl, st);
    if (f.has_post)
      return dispatch_fdm_mode<false, true>(g, f, a, tl, st);
    return dispatch_fdm_mode<false, false>(g, f, a, tl, st);
  }
#ifdef ASM_STAMP
  // workgroup trace of the diagnostic builds (tools/wgtrace_run.py); not part of the C ABI
  extern "C" int
  asm_debug_wgtrace(unsigned long long *out, int max_rows, int reset)
  {
    (void)hipDeviceSynchronize();
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgtrace_n), sizeof(n));
    if (n > 16384u)
      n = 16384u;
    if ((int)n > max_rows)
      n = (unsigned)max_rows;
    if (n > 0 && out != nullptr)
      (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgtrace), (size_t)n * 4 * sizeof(unsigned long long));
    if (reset)
      {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgtrace_n), &z, sizeof(z));
      }
    return (int)n;
  }
  void
  tiled_debug_stamps(unsigned long long out[32], bool reset)
  {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 32 * sizeof(unsigned long long));
    if (reset)
      {
        unsigned long long z[32] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
      }
  }
#endif
} // namespace asmamd
