// Wave-pair sweep for degree 4 on tensor-product meshes (gfx950): the FDM additive-Schwarz preconditioner with cell
// patches (n_overlap = 1) and the Laplace operator on box cells, both as the "Kronecker-diagonal" cell operator
//   dst = update( sum_cells R^T (Bz x By x Bx) diag(D) (Fz x Fy x Fx)^T R src )
// of the tables in DevKron (grid.h).  It replaces, like the tiled kernels (kernels_tiled.hip), the gather /
// apply_inverse / scatter-add of ASPoissonPreconditioner::vmult_internal (include/matrix_free.h:1023-1062), the cell loop
// of LaplaceOperatorMatrixFree::vmult (include/operator.h:1367-1430) and the pre/post lambdas of PreconditionChebyshev.
//
// Work decomposition (round 4; measurements behind it: profiles/r04_pair_design.txt)
//   * ONE LANE PER CELL: a workgroup = two waves that share a tile of 8 x 8 cells (lane l of both waves = cell l) and
//     march along z over a chunk of cell layers; the tile owns the DoFs of 7 x 7 cells (first column / row = halo,
//     computed for their contributions only), every DoF is written exactly once by the lane that owns it: no atomics.
//     With a whole cell per lane all three sweep directions are register FMA chains: no transposes.
//   * A cell needs 125 values at once, more than a lane's arithmetic registers hold beside anything else.  So per cell
//     layer wave r transforms two of the four NEW DoF planes in x-y (planes 1 + 2r, 2 + 2r; plane 0 is plane 4 of the
//     layer below: its transform is kept), both write their planes to an LDS scratchpad (lane-strided addresses:
//     conflict-free, no address arithmetic), each wave takes 13 / 12 of the 25 z columns through the z sweeps and the
//     diagonal in place in the scratchpad, and transforms two of the four output planes back (planes 2r, 2r + 1; the
//     contributions of two consecutive cells of a column to their common plane are added BEFORE the back transform:
//     4 instead of 5 plane transforms each way).  Two barriers of two waves per layer, each wave alone on its SIMD.
//   * ... or FOUR waves per tile (NW = 4, every mode but the three-term update): wave r transforms plane 1 + r, takes
//     7 / 6 / 6 / 6 z columns and transforms output plane r back.  A wave then needs half the registers (one plane and one
//     operand plane in flight: 128 + 128), two waves share a SIMD, and one wave's scratchpad writes, requests and waits
//     overlap the other's arithmetic: alone on its SIMD a wave executes instructions 47 % of the time (SQ counters,
//     profiles/r04_pair_design.txt).
//   * Nothing is parked in registers, so the accumulation registers are free: they take the NEXT layer's source planes
//     and update operands, requested a whole layer ahead by plain global loads into FIXED registers the compiler never
//     sees (one wave per SIMD has no other wave to hide a miss behind).  Audit of every build: no v_accvgpr_* outside
//     the asm statements above a91 / a27 (three-term update) / a17 (four waves), no scratch, no flat instruction
//     (tools/audit_pair.sh).
//   * Merges: x by a DPP row shift, y by ds_bpermute, z in the transformed space (carry of 13 values per lane).
//   * The cells c = nc of a direction whose last DoF plane belongs to the block do not exist but OWN that plane (their
//     column / row 0): they run along with a zero diagonal and store what they receive from their neighbours.
//   * Constrained DoFs are neither read (the tables zero their rows) nor written by the march; further workgroups of the same
//     launch set them (constrained_dofs).
//   * Tiles: 8 x 8 cells, and strips of 4 x 16 / 16 x 4 cells where a last tile column / row would own three cells or less
//     (PairTiling, tile_geom).  z chunks and the form (two / four waves) are measured by the context (context.hip: tune_pair).
//   * Two-phase launches for the overlapped ghost exchange of a partitioned run: pair_phase_order.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

// Stores of the sweep's result go through the L2 (write-back), not past it: a tile's row of 7 cells = 224 B starts in the
// middle of a 64-byte sector in every other tile, and the two halves of such a sector come from two workgroups - streaming
// stores write them to memory separately (320 MB written per FDM sweep at 80^3 cells instead of 286 MB, 265 MB algorithmic;
// 1 % of the sweep).  -DASM_PAIR_NT_STORES: the streaming form.
#ifdef ASM_PAIR_NT_STORES
#  define PAIR_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#  define PAIR_STORE(v, p) (*(p) = (v))
#endif

namespace asmamd
{
  namespace
  {
    constexpr int P = 4, N1 = 5, NE = 25;
    constexpr int TX = 8, TY = 8, OX = TX - 1, OY = TY - 1;
    // z columns of wave 0 of NW waves (0..12 of two, 0..6 of four); the others take one less each: 13..24, or 7..12, 13..18, 19..24
    constexpr int
    z_columns(const int nw)
    {
      return nw == 2 ? 13 : 7;
    }

    // LDS of a workgroup (doubles)
    constexpr int L_X    = P * NE * 64;                                // scratchpad: slot (q, e) of lane l at (q * 25 + e) * 64 + l
    constexpr int L_D    = KRON_MAX_CLS * KRON_MAX_CLS * KRON_DIAG_STRIDE; // diagonals of the current z class, [cly][clx]
    constexpr int L_ZERO = 72;                                         // the diagonal of cells that are not computed: 13 columns of zeros
    constexpr int L_M    = 3 * KRON_MAX_CLS * KRON_MAT_STRIDE;         // all matrices
    constexpr int L_TOTAL = L_X + L_D + L_ZERO + L_M;
    static_assert(2 * L_TOTAL * sizeof(double) <= 160 * 1024, "two workgroups per compute unit");

    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef dbl2   dbl2_u __attribute__((aligned(8)));
    // LDS pointers carry their address space in the type: a generic pointer would compile to flat loads, which count in
    // vmcnt as well and complete out of order - the kernel's wait counts assume that no flat instruction exists (audit)
    typedef __attribute__((address_space(3))) double       *lds_ptr;
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    typedef __attribute__((address_space(1))) double       *glb_ptr;
    typedef __attribute__((address_space(1))) dbl2_u       *glb_dbl2_ptr;

    // Tiles: ntx x nty "main" tiles of 8 x 8 cells (7 x 7 owned), then - where the last tile column (row) of such a tiling
    // would own three cell columns (rows) or less - n_rs strip tiles of 4 x 16 cells (3 x 15 owned) stacked along the block's
    // right edge and n_ts strip tiles of 16 x 4 cells along its upper edge instead of it: 81 = 8 + 10 x 7 + 3 cell columns
    // (80 cells and the one behind them) take 11 tile columns and 6 strips, not 12 columns whose last one is 3/7 full
    // (133 instead of 144 tiles at 80^3, 342 instead of 361 at 128 x 128).  The shape is a run-time matter of the lane -> cell
    // map only; tile index t = main tiles (x fastest), right strips (bottom up), upper strips (left to right).
    struct PairTiling
    {
      int ntx, nty, nzc, zlen;
      int x_first, y_first; // first cell of tile 0: -1 where the block has a cell below its first one, else 0
      int n_rs, n_ts;       // strip tiles at the right / upper edge
      int xs0, ys0;         // ... whose halo column / row is this cell (= the last cell the main tiles own)
      __host__ __device__ int n_tiles() const { return ntx * nty + n_rs + n_ts; }
      // The Dirichlet planes, which the march neither reads nor writes, are set by n_face further workgroups at the end of
      // the launch (identity rows: the fused update of the source value).  They start as compute units fall free in the
      // launch's tail and cost next to nothing there; as a kernel of their own behind every sweep (vec_constrained*,
      // 64-byte sectors for 8-byte values on the x faces) they took 24 us = 8 % of a sweep at 80^3.
      int       n_face;
      long long face_dofs; // DoFs of all Dirichlet planes (stored extent of the two tangential directions)
    };

    struct TileGeom
    {
      int  X0, Y0;     // cell of lane 0
      int  lg_tx;      // log2 of the tile's width in cells: 3 (8 x 8), 2 (4 x 16), 4 (16 x 4)
      bool lo_x, lo_y; // the tile starts at the block's first cell at a physical boundary: it owns its first column / row
      int  x_max;      // cells beyond this column belong to another tile (upper strips next to the right strips)
    };
    __host__ __device__ __forceinline__ TileGeom
    tile_geom(const PairTiling &tl, const int t)
    {
      TileGeom  q;
      const int n_main = tl.ntx * tl.nty;
      q.x_max          = 1 << 30;
      if (t < n_main)
        {
          const int tx = t % tl.ntx, ty = t / tl.ntx;
          q.X0    = tx * OX + tl.x_first;
          q.Y0    = ty * OY + tl.y_first;
          q.lg_tx = 3;
          q.lo_x  = tx == 0 && tl.x_first == 0;
          q.lo_y  = ty == 0 && tl.y_first == 0;
        }
      else if (t < n_main + tl.n_rs)
        {
          const int s = t - n_main;
          q.X0    = tl.xs0;
          q.Y0    = s * 15 + tl.y_first;
          q.lg_tx = 2;
          q.lo_x  = false;
          q.lo_y  = s == 0 && tl.y_first == 0;
        }
      else
        {
          const int s = t - n_main - tl.n_rs;
          q.X0    = s * 15 + tl.x_first;
          q.Y0    = tl.ys0;
          q.lg_tx = 4;
          q.lo_x  = s == 0 && tl.x_first == 0;
          q.lo_y  = false;
          if (tl.n_rs > 0)
            q.x_max = tl.xs0;
        }
      return q;
    }

    __device__ __forceinline__ bool
    cell_computable(const DevGrid &g, const int d, const int c)
    {
      if (c < -1 || c >= g.nc[d])
        return false;
      if (c >= 0)
        return true;
      return g.wrap[d] || g.ext_lo[d] < 0; // periodic image or halo cell of the lower rank
    }
    // the cell behind the block's last one: it does not exist but owns the block's last DoF plane
    __host__ __device__ __forceinline__ bool
    has_phantom(const DevGrid &g, const int d)
    {
      return !g.wrap[d] && g.own_hi[d] == g.nc[d] * P;
    }

    template <typename T>
    __device__ __forceinline__ T
    uniform_load(const T *table, const int idx)
    {
      return ((const __attribute__((address_space(4))) T *)table)[__builtin_amdgcn_readfirstlane(idx)];
    }

    // ---- coefficient registers, read from the LDS copy of the tables (same address in all lanes of a class: broadcast)
    // even-odd form: E[a][i] = q[i][a] (a, i < 3), O[a][i] = q[i][3 + a] (a, i < 2)
    struct EoCoef
    {
      double e[9], o[4];
      __device__ __forceinline__ explicit EoCoef(lds_cptr m)
      {
#pragma unroll
        for (int k = 0; k < 9; ++k)
          e[k] = m[k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = m[9 + k];
      }
      __device__ __forceinline__ double E(const int a, const int i) const { return e[a * 3 + i]; }
      __device__ __forceinline__ double O(const int a, const int i) const { return o[a * 2 + i]; }
    };
    // full matrix (cells at a boundary have no even-odd form): five rows of six in LDS, output r = row r . input, for the
    // forward sweep the rows of F^T, for the backward sweep the rows of B.  The pointer is the lane's own (its cell's class)
    // or wave-uniform (z): no lane masks, no scalar loads.  The rows are read again for every line - 25 coefficients held
    // over a sweep would push the compiler into the accumulation registers this kernel keeps for itself.
    struct FullCoef
    {
      lds_cptr q;
      __device__ __forceinline__ explicit FullCoef(lds_cptr m)
        : q(m)
      {}
    };
    // ... and the same 25 numbers held in registers over one sweep of a plane (five lines): the rows' LDS latency is paid
    // once per sweep instead of once per line (410 cycles per line in a tile at a face, stage timers of round 4)
    struct FullRegs
    {
      double r[N1][N1];
      __device__ __forceinline__ explicit FullRegs(lds_cptr m)
      {
#pragma unroll
        for (int k = 0; k < N1; ++k)
          {
            const dbl2 q01 = *(const __attribute__((address_space(3))) dbl2 *)(m + k * 6), q23 = *(const __attribute__((address_space(3))) dbl2 *)(m + k * 6 + 2);
            r[k][0]        = q01.x;
            r[k][1]        = q01.y;
            r[k][2]        = q23.x;
            r[k][3]        = q23.y;
            r[k][4]        = m[k * 6 + 4];
          }
      }
    };

    // The same rows as scalar operands (scalar loads from the table in global memory), for ONE class at a time: no vector
    // registers at all.  The three-term update's kernel takes this form under lane masks, class by class: its second
    // operand leaves the compiler 28 accumulation registers, the LDS form above makes it spill 92 values.
    struct FullCoefS
    {
      const __attribute__((address_space(4))) double *q;
      __device__ __forceinline__ explicit FullCoefS(const double *m)
        : q((const __attribute__((address_space(4))) double *)m)
      {}
    };

    // FWD: out[a] = sum_i q[i][a] in[i];  else out[i] = sum_a q[i][a] in[a]
    template <bool FWD>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const EoCoef &q)
    {
      if (FWD)
        {
          const double e0 = u0 + u4, e1 = u1 + u3, o0 = u0 - u4, o1 = u1 - u3, m = u2;
          u0 = fma(q.E(0, 2), m, fma(q.E(0, 1), e1, q.E(0, 0) * e0));
          u1 = fma(q.E(1, 2), m, fma(q.E(1, 1), e1, q.E(1, 0) * e0));
          u2 = fma(q.E(2, 2), m, fma(q.E(2, 1), e1, q.E(2, 0) * e0));
          u3 = fma(q.O(0, 1), o1, q.O(0, 0) * o0);
          u4 = fma(q.O(1, 1), o1, q.O(1, 0) * o0);
        }
      else
        {
          const double E0 = fma(q.E(2, 0), u2, fma(q.E(1, 0), u1, q.E(0, 0) * u0));
          const double E1 = fma(q.E(2, 1), u2, fma(q.E(1, 1), u1, q.E(0, 1) * u0));
          const double E2 = fma(q.E(2, 2), u2, fma(q.E(1, 2), u1, q.E(0, 2) * u0));
          const double O0 = fma(q.O(1, 0), u4, q.O(0, 0) * u3);
          const double O1 = fma(q.O(1, 1), u4, q.O(0, 1) * u3);
          u0              = E0 + O0;
          u4              = E0 - O0;
          u1              = E1 + O1;
          u3              = E1 - O1;
          u2              = E2;
        }
    }
    template <bool FWD>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const FullCoef &c)
    {
      // (a pointer the compiler cannot trace and that exists only once this line's input does: the reads of this line are
      //  neither merged with those of the line before nor issued lines ahead, which would cost 30 registers per line)
      lds_cptr q = c.q;
      asm volatile("" : "+v"(q) : "v"(u0));
      double o[N1];
#pragma unroll
      for (int r = 0; r < N1; ++r)
        {
          const dbl2   q01 = *(const __attribute__((address_space(3))) dbl2 *)(q + r * 6), q23 = *(const __attribute__((address_space(3))) dbl2 *)(q + r * 6 + 2);
          const double q4  = q[r * 6 + 4];
          o[r]             = fma(q4, u4, fma(q23.y, u3, fma(q23.x, u2, fma(q01.y, u1, q01.x * u0))));
        }
      u0 = o[0];
      u1 = o[1];
      u2 = o[2];
      u3 = o[3];
      u4 = o[4];
    }
    template <bool FWD>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const FullRegs &c)
    {
      double o[N1];
#pragma unroll
      for (int r = 0; r < N1; ++r)
        o[r] = fma(c.r[r][4], u4, fma(c.r[r][3], u3, fma(c.r[r][2], u2, fma(c.r[r][1], u1, c.r[r][0] * u0))));
      u0 = o[0];
      u1 = o[1];
      u2 = o[2];
      u3 = o[3];
      u4 = o[4];
    }
    template <bool FWD>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const FullCoefS &c)
    {
      const __attribute__((address_space(4))) double *q = c.q;
      double                                          o[N1];
#pragma unroll
      for (int r = 0; r < N1; ++r)
        o[r] = fma(q[r * 6 + 4], u4, fma(q[r * 6 + 3], u3, fma(q[r * 6 + 2], u2, fma(q[r * 6 + 1], u1, q[r * 6 + 0] * u0))));
      u0 = o[0];
      u1 = o[1];
      u2 = o[2];
      u3 = o[3];
      u4 = o[4];
    }
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_x(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int j = 0; j < N1; ++j)
        line5<FWD>(v[j * 5 + 0], v[j * 5 + 1], v[j * 5 + 2], v[j * 5 + 3], v[j * 5 + 4], q);
    }
    template <bool FWD, typename Q>
    __device__ __forceinline__ void
    sweep_y(double (&v)[NE], const Q &q)
    {
#pragma unroll
      for (int i = 0; i < N1; ++i)
        line5<FWD>(v[0 * 5 + i], v[1 * 5 + i], v[2 * 5 + i], v[3 * 5 + i], v[4 * 5 + i], q);
    }

    // Per-lane conditions live as bits of ONE vector register and are tested where they are used: as booleans the
    // compiler would keep each of them as a 64-bit lane mask in two scalar registers over the whole march, and the kernel's
    // scalar registers do not suffice for that (they went to vector lanes and back: ~200 moves per layer)
    enum LaneFlag : unsigned
    {
      F_CELL   = 1u,    // the cell is computed
      F_OWNER  = 2u,    // the lane stores DoFs (its cell, or the cell that does not exist but owns the last plane)
      F_PH_X   = 4u,    // ... of which only column 0 exists
      F_PH_Y   = 8u,    // ... of which only row 0 exists
      F_XD0    = 16u,   // column 0 is a Dirichlet plane: not stored
      F_YD0    = 32u,   // row 0 is a Dirichlet plane: not stored
      F_LEFT   = 64u,   // the lane to the left belongs to the tile
      F_BELOW  = 128u,  // the lane below belongs to the tile
      F_NSX    = 256u,  // the cell's x class has no even-odd form
      F_NSY    = 512u,  // the cell's y class has no even-odd form
      F_SM_X0  = 1024u, // source DoFs read as zero (Laplace operator at Dirichlet faces): column 0, column 4, row 0, row 4
      F_SM_X4  = 2048u,
      F_SM_Y0  = 4096u,
      F_SM_Y4  = 8192u
    };
    struct Lane
    {
      unsigned flags;
      // source: byte offset of DoF row 0 and row 4 of the cell inside a plane (cells not computed: cell 0), rows 1..3 follow
      // row 0 at the row stride (row 4 may lie behind a periodic seam), and the step from column 0 to column 4 (seam)
      unsigned srow0, srow4, sd4;
      // operands / dst: byte offset of row 0 of the DoFs the lane owns and the step to the next row (0: only row 0 exists)
      unsigned orow0, ostep;
      unsigned sstep; // the row stride in bytes (wave-uniform, kept with the rest)
      int      tw;    // cells per tile row (8, 4 or 16): the lane below is tw lanes down
      __device__ __forceinline__ bool is(const unsigned f) const { return (flags & f) != 0u; }
      // (the tests are made again in every layer instead of being kept as lane masks over the march)
      __device__ __forceinline__ void refresh() { asm volatile("" : "+v"(flags)); }
      // the offsets are formed where they are used, from values the compiler cannot trace: it would otherwise keep all
      // rows' offsets, their seam variants and zero-extended copies over the march - registers a wave of four per tile lacks
      template <int J>
      __device__ __forceinline__ unsigned srow() const
      {
        unsigned b = J == 4 ? srow4 : srow0;
        asm volatile("" : "+v"(b));
        return J == 4 || J == 0 ? b : b + (unsigned)J * sstep;
      }
      template <int J>
      __device__ __forceinline__ unsigned orow() const
      {
        unsigned b = orow0;
        asm volatile("" : "+v"(b));
        return J == 0 ? b : b + (unsigned)J * ostep;
      }
    };

    // The x and y matrices of a lane (its cell's classes) in LDS: [F^T 30][B 30][F even-odd 16][B even-odd 16].
    // any_ns* (wave-uniform): some cell of the wave has a class without an even-odd form in that direction.  LDSFULL: the
    // whole wave then takes the full matrices there (25 instead of 17 operations per line, every lane with its own
    // class's rows from LDS); otherwise (three-term kernel) the lanes of such classes are redone class by class under
    // lane masks with scalar operands.
    struct XYMatrices
    {
      lds_cptr      mx, my;
      bool          any_nsx, any_nsy;
      int           clx, cly;
      const double *gx, *gy; // the tables of the x / y classes in global memory: [class][KRON_MAT_STRIDE]
    };
    // one sweep direction of a plane
    template <bool FWD, bool XDIR, bool LDSFULL, bool FULLREGS>
    __device__ __forceinline__ void
    sweep_dir(double (&v)[NE], const XYMatrices &m, const Lane &c)
    {
      if (LDSFULL && !(XDIR ? m.any_nsx : m.any_nsy))
        {
          const EoCoef q((XDIR ? m.mx : m.my) + (FWD ? 60 : 76));
          if (XDIR)
            sweep_x<FWD>(v, q);
          else
            sweep_y<FWD>(v, q);
        }
      else if (LDSFULL && FULLREGS)
        {
          const FullRegs q((XDIR ? m.mx : m.my) + (FWD ? 0 : 30));
          if (XDIR)
            sweep_x<FWD>(v, q);
          else
            sweep_y<FWD>(v, q);
        }
      else if (LDSFULL)
        {
          // (four waves per tile: a wave has no 50 registers for the matrix; its rows come from LDS line by line)
          const FullCoef q((XDIR ? m.mx : m.my) + (FWD ? 0 : 30));
          if (XDIR)
            sweep_x<FWD>(v, q);
          else
            sweep_y<FWD>(v, q);
        }
      else
        {
          const bool ns = c.is(XDIR ? F_NSX : F_NSY);
          if (!ns)
            {
              const EoCoef q((XDIR ? m.mx : m.my) + (FWD ? 60 : 76));
              if (XDIR)
                sweep_x<FWD>(v, q);
              else
                sweep_y<FWD>(v, q);
            }
          if (!(XDIR ? m.any_nsx : m.any_nsy))
            return;
          // (wave-uniform loop over the classes without an even-odd form that occur in the wave: one, rarely two)
          const int          cls  = XDIR ? m.clx : m.cly;
          unsigned long long todo = __ballot(ns);
          while (todo != 0ull)
            {
              const int  c0   = __builtin_amdgcn_readlane(cls, __ffsll((long long)todo) - 1);
              const bool mine = ns && cls == c0;
              if (mine)
                {
                  const FullCoefS q((XDIR ? m.gx : m.gy) + (long long)c0 * KRON_MAT_STRIDE + (FWD ? 0 : 30));
                  if (XDIR)
                    sweep_x<FWD>(v, q);
                  else
                    sweep_y<FWD>(v, q);
                }
              todo &= ~__ballot(mine);
            }
        }
    }
    // x-y transform of one plane v[j * 5 + i]: forward x then y with F, backward y then x with B
    template <bool FWD, bool LDSFULL, bool FULLREGS>
    __device__ __forceinline__ void
    plane2d(double (&v)[NE], const XYMatrices &m, const Lane &c)
    {
      if (FWD)
        {
          sweep_dir<true, true, LDSFULL, FULLREGS>(v, m, c);
          sweep_dir<true, false, LDSFULL, FULLREGS>(v, m, c);
        }
      else
        {
          sweep_dir<false, false, LDSFULL, FULLREGS>(v, m, c);
          sweep_dir<false, true, LDSFULL, FULLREGS>(v, m, c);
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // The next layer's source planes and update operands wait in FIXED accumulation registers which the compiler never
    // sees; it takes a0.. upwards if it has to spill (audit: its v_accvgpr_* stay below A_BASE, no scratch):
    //   a[A_S0 .. +49], a[A_S1 .. +49]   the two source planes of this wave: row j at +10 j: x 0..1, x 2..3, x 4
    //   a[A_XA0 .. +31], a[A_XA1 .. +31] first update operand of its two output planes: row j at +8 j
    //   a[A_XB0 .. +31], a[A_XB1 .. +31] second update operand (three-term update only)
    //   a[A_P0 .. +49]                   the prologue's plane 0: the registers of the operands requested after it
    // The three-term update leaves the compiler a0..a27, the other modes a0..a91.
    // ---------------------------------------------------------------------------------------------------------------
    template <bool NEED_B, int NW>
    struct ARegs
    {
      static constexpr int BASE = NEED_B ? 28 : 92;
      static constexpr int S0 = BASE, S1 = BASE + 50, XA0 = BASE + 100, XA1 = BASE + 132, XB0 = NEED_B ? BASE + 164 : 0, XB1 = NEED_B ? BASE + 196 : 0;
      static constexpr int P0 = NEED_B ? XB0 : XA0;
      static constexpr int T0 = 0, CY = 0; // (two waves per tile: in arithmetic registers)
      static_assert((NEED_B ? XB1 : XA1) + 32 == 256 && P0 + 50 <= 256, "accumulation register map");
    };
    // four waves per tile: one source plane and one operand plane per wave, in the 128 accumulation registers two waves
    // per SIMD leave each of them; the prologue's plane goes through the source plane's registers.  The compiler has a0..a17.
    template <>
    struct ARegs<false, 4>
    {
      static constexpr int BASE = 18;
      // (the z sweeps' carried values - plane 4 of the layer below in transformed space, and the contributions to plane 0 of
      //  the layer above - of this wave's 7 columns live here too: 128 arithmetic registers do not hold them over the march)
      static constexpr int T0 = BASE, CY = BASE + 14;
      static constexpr int S0 = BASE + 28, S1 = S0, XA0 = S0 + 50, XA1 = XA0, XB0 = 0, XB1 = 0;
      static constexpr int P0 = S0;
      static_assert(XA0 + 32 == 128, "accumulation register map");
    };
#define ASM_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
    // (makes the kernel's register count cover the fixed registers; the statement itself does nothing)
    template <int NW>
    __device__ __forceinline__ void
    areg_reserve()
    {
      if (NW == 2)
        asm volatile("" ::: ASM_A10(10), ASM_A10(11), ASM_A10(12), ASM_A10(13), ASM_A10(14), ASM_A10(15), ASM_A10(16), ASM_A10(17), ASM_A10(18),
                     ASM_A10(19), ASM_A10(20), ASM_A10(21), ASM_A10(22), ASM_A10(23), ASM_A10(24), "a250", "a251", "a252", "a253", "a254", "a255");
      else
        asm volatile("" ::: ASM_A10(5), ASM_A10(6), ASM_A10(7), ASM_A10(8), ASM_A10(9), ASM_A10(10), ASM_A10(11), "a120", "a121", "a122", "a123", "a124",
                     "a125", "a126", "a127");
    }
#undef ASM_A10

    // five values of DoF row J: x 0..3 from the row's offset, x 4 from its own (periodic seam)
    template <int A0, int J>
    __device__ __forceinline__ void
    plane_request_row(const double *plane, const unsigned o, const unsigned d4)
    {
      const unsigned o4 = o + d4;
      // (s_nop: the base may just have come back from a vector lane - v_readlane writes the scalar register, and a vector
      //  memory instruction that reads a scalar register a VALU instruction wrote needs 5 wait states in between; the
      //  compiler pads for that hazard everywhere except in front of an asm statement, whose content it does not know)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 a[%3:%4], %0, %2\n\t"
                   "global_load_dwordx4 a[%5:%6], %0, %2 offset:16\n\t"
                   "global_load_dwordx2 a[%7:%8], %1, %2" ::"v"(o),
                   "v"(o4), "s"(plane), "n"(A0 + 10 * J), "n"(A0 + 10 * J + 3), "n"(A0 + 10 * J + 4), "n"(A0 + 10 * J + 7), "n"(A0 + 10 * J + 8),
                   "n"(A0 + 10 * J + 9)
                   : "memory");
    }
    template <int A0, int J>
    __device__ __forceinline__ void
    operand_request_row(const double *plane, const unsigned o)
    {
      asm volatile("s_nop 4\n\t" // (as above)
                   "global_load_dwordx4 a[%2:%3], %0, %1\n\t"
                   "global_load_dwordx4 a[%4:%5], %0, %1 offset:16" ::"v"(o),
                   "s"(plane), "n"(A0 + 8 * J), "n"(A0 + 8 * J + 3), "n"(A0 + 8 * J + 4), "n"(A0 + 8 * J + 7)
                   : "memory");
    }
    // everything but the N youngest vector memory operations of the wave has completed
    template <int N>
    __device__ __forceinline__ void
    vm_wait()
    {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    }
    template <int A>
    __device__ __forceinline__ double
    areg_double()
    {
      int lo, hi;
      asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]" : "=v"(lo), "=v"(hi) : "n"(A), "n"(A + 1));
      return __hiloint2double(hi, lo);
    }
    template <int A>
    __device__ __forceinline__ void
    areg_set(const double x)
    {
      asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(__double2loint(x)), "v"(__double2hiint(x)), "n"(A), "n"(A + 1));
    }
    // f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>): loops whose index names a register
    template <typename F, int... I>
    __device__ __forceinline__ void
    static_for_impl(F &&f, std::integer_sequence<int, I...>)
    {
      (f(std::integral_constant<int, I>{}), ...);
    }
    template <int N, typename F>
    __device__ __forceinline__ void
    static_for(F &&f)
    {
      static_for_impl(f, std::make_integer_sequence<int, N>());
    }
    template <int A0, int... E>
    __device__ __forceinline__ void
    aregs_to(double *v, std::integer_sequence<int, E...>)
    {
      ((v[E] = areg_double<A0 + 2 * E>()), ...); // element E of a block of doubles at a[A0 + 2 E]
    }
    template <int A0>
    __device__ __forceinline__ void
    plane_unpack(double (&v)[NE])
    {
      aregs_to<A0 + 0>(&v[0], std::make_integer_sequence<int, 5>());
      aregs_to<A0 + 10>(&v[5], std::make_integer_sequence<int, 5>());
      aregs_to<A0 + 20>(&v[10], std::make_integer_sequence<int, 5>());
      aregs_to<A0 + 30>(&v[15], std::make_integer_sequence<int, 5>());
      aregs_to<A0 + 40>(&v[20], std::make_integer_sequence<int, 5>());
    }
    // ---------------------------------------------------------------------------------------------------------------
    template <int A0>
    __device__ __forceinline__ void
    plane_request(const double *plane, const Lane &c)
    {
      plane_request_row<A0, 0>(plane, c.srow<0>(), c.sd4);
      plane_request_row<A0, 1>(plane, c.srow<1>(), c.sd4);
      plane_request_row<A0, 2>(plane, c.srow<2>(), c.sd4);
      plane_request_row<A0, 3>(plane, c.srow<3>(), c.sd4);
      plane_request_row<A0, 4>(plane, c.srow<4>(), c.sd4);
    }
    template <int A0>
    __device__ __forceinline__ void
    operand_request(const double *plane, const Lane &c)
    {
      operand_request_row<A0, 0>(plane, c.orow<0>());
      operand_request_row<A0, 1>(plane, c.orow<1>());
      operand_request_row<A0, 2>(plane, c.orow<2>());
      operand_request_row<A0, 3>(plane, c.orow<3>());
    }

    // constrained DoFs of a source plane read as zero (whole plane: a Dirichlet plane in z)
    __device__ __forceinline__ void
    mask_source(double (&v)[NE], const Lane &c, const bool plane_constrained)
    {
      if (plane_constrained)
        {
#pragma unroll
          for (int e = 0; e < NE; ++e)
            v[e] = 0.;
        }
      else if (c.is(F_SM_X0 | F_SM_X4 | F_SM_Y0 | F_SM_Y4))
        {
#pragma unroll
          for (int j = 0; j < N1; ++j)
#pragma unroll
            for (int i = 0; i < N1; ++i)
              if ((i == 0 && c.is(F_SM_X0)) || (i == 4 && c.is(F_SM_X4)) || (j == 0 && c.is(F_SM_Y0)) || (j == 4 && c.is(F_SM_Y4)))
                v[j * 5 + i] = 0.;
        }
    }

    // one tenth of the requests of the next layer's two planes (step 0..9 = plane step / 5, row step % 5; 0..4 with one
    // plane per wave): spread over
    // the z columns instead of issued in one burst, which the vector memory path accepts at 60 cycles per request
    // (stage timers, profiles/r04_pair_stamps.txt)
    template <int AS0, int AS1, int PPW>
    __device__ __forceinline__ void
    plane_request_step(const int step, const double *p0, const double *p1, const Lane &c)
    {
      if (PPW == 1 && step >= 5)
        return;
      switch (step)
        {
          case 0: plane_request_row<AS0, 0>(p0, c.srow<0>(), c.sd4); break;
          case 1: plane_request_row<AS0, 1>(p0, c.srow<1>(), c.sd4); break;
          case 2: plane_request_row<AS0, 2>(p0, c.srow<2>(), c.sd4); break;
          case 3: plane_request_row<AS0, 3>(p0, c.srow<3>(), c.sd4); break;
          case 4: plane_request_row<AS0, 4>(p0, c.srow<4>(), c.sd4); break;
          case 5: plane_request_row<AS1, 0>(p1, c.srow<0>(), c.sd4); break;
          case 6: plane_request_row<AS1, 1>(p1, c.srow<1>(), c.sd4); break;
          case 7: plane_request_row<AS1, 2>(p1, c.srow<2>(), c.sd4); break;
          case 8: plane_request_row<AS1, 3>(p1, c.srow<3>(), c.sd4); break;
          case 9: plane_request_row<AS1, 4>(p1, c.srow<4>(), c.sd4); break;
          default: break;
        }
    }

    template <int MODE>
    struct ModeTraits
    {
      static constexpr bool need_a = (MODE == OP_RESIDUAL || MODE == PC_STEP1 || MODE == PC_UPDATE);
      static constexpr bool need_b = (MODE == PC_UPDATE);
      static constexpr int  n_operand_requests = (need_a ? 8 : 0) + (need_b ? 8 : 0); // per output plane
    };
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

    // value of lane - 1 inside a row of 16 lanes (DPP row_shr:1; the first lane of a row reads 0)
    __device__ __forceinline__ double
    lane_left(const double x)
    {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
      return __hiloint2double(hi, lo);
    }

    // fused update and store of DoF row J of a plane: the operands come out of the accumulation registers row by row
    template <int MODE, int J, int AXA, int AXB>
    __device__ __forceinline__ void
    emit_row(const TiledArgs &a, const Lane &c, const glb_ptr dplane, const double (&o)[NE])
    {
      using MT = ModeTraits<MODE>;
      if (J == 0 ? !c.is(F_YD0) : !c.is(F_PH_Y))
        {
          double xa[P] = {0., 0., 0., 0.}, xb[P] = {0., 0., 0., 0.};
          if (MT::need_a)
            aregs_to<AXA + 8 * J>(xa, std::make_integer_sequence<int, P>());
          if (MT::need_b)
            aregs_to<AXB + 8 * J>(xb, std::make_integer_sequence<int, P>());
          // (the offset's zero extension must happen here, next to the stores: extended ahead of the loop - as the compiler
          //  likes to - it costs a register pair per row and a 64-bit vector add instead of the scalar-base address form)
          const unsigned off = c.orow<J>();
          const glb_ptr d = (glb_ptr)((__attribute__((address_space(1))) char *)dplane + off);
          dbl2          lo, hi;
          lo.x = fused_update<MODE>(a, o[J * 5 + 0], xa[0], xb[0]);
          lo.y = fused_update<MODE>(a, o[J * 5 + 1], xa[1], xb[1]);
          hi.x = fused_update<MODE>(a, o[J * 5 + 2], xa[2], xb[2]);
          hi.y = fused_update<MODE>(a, o[J * 5 + 3], xa[3], xb[3]);
          if (c.is(F_PH_X))
            {
              if (!c.is(F_XD0))
                PAIR_STORE(lo.x, d);
            }
          else if (c.is(F_XD0))
            {
              PAIR_STORE(lo.y, d + 1);
              PAIR_STORE(hi, (glb_dbl2_ptr)(d + 2));
            }
          else
            {
              PAIR_STORE(lo, (glb_dbl2_ptr)d);
              PAIR_STORE(hi, (glb_dbl2_ptr)(d + 2));
            }
        }
    }

    // merges in x and y, fused update and store of the DoFs of one plane (at dst + plane offset) the lane owns;
    // o[j * 5 + i] is consumed.  Cells that are not computed hold exact zeros (zero diagonal).
    template <int MODE, int AXA, int AXB>
    __device__ __forceinline__ void
    emit_plane(const TiledArgs &a, const Lane &c, double *__restrict__ dplane, double (&o)[NE], const bool plane_constrained)
    {
      // x: column 4 of the left cell -> column 0
#pragma unroll
      for (int j = 0; j < N1; ++j)
        {
          const double r = lane_left(o[j * 5 + 4]);
          if (c.is(F_LEFT))
            o[j * 5 + 0] += r;
        }
      // y: row 4 of the cell below -> row 0
      double rb[P];
#pragma unroll
      for (int i = 0; i < P; ++i)
        rb[i] = __shfl_up(o[4 * 5 + i], c.tw, 64);
      if (c.is(F_BELOW))
        {
#pragma unroll
          for (int i = 0; i < P; ++i)
            o[0 * 5 + i] += rb[i];
        }
      if (c.is(F_OWNER) && !plane_constrained)
        {
          // (scalar base + the row's 32-bit offset: see Lane::refresh; the pointer keeps its address space through the asm)
          glb_ptr dp = (glb_ptr)dplane;
          asm volatile("" : "+s"(dp));
          emit_row<MODE, 0, AXA, AXB>(a, c, dp, o);
          emit_row<MODE, 1, AXA, AXB>(a, c, dp, o);
          emit_row<MODE, 2, AXA, AXB>(a, c, dp, o);
          emit_row<MODE, 3, AXA, AXB>(a, c, dp, o);
        }
    }

    // storage offset (doubles) of DoF plane K
    __device__ __forceinline__ long long
    plane_offset(const DevGrid &g, const int K)
    {
      // (wave-uniform, but the compiler computes it on the vector unit when it runs short of scalar registers - and would
      //  then hand the asm requests a vector register pair as their scalar base)
      const long long o = g.store_index_near(2, K) * g.stride[2];
      return ((long long)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)o);
    }

#ifdef ASM_PAIR_STAMP
    // diagnostic build: cycles (s_memtime) one interior wave spends per stage, summed over its layers
    __device__ unsigned long long g_pair_stamps[6][24]; // [mode]
#  define PSTAMP(i)                                                    \
    do                                                                 \
      {                                                                \
        if (stamp_on)                                                  \
          {                                                            \
            const unsigned long long t = __builtin_amdgcn_s_memtime(); \
            stamp_acc[i] += t - stamp_t;                               \
            stamp_t = t;                                               \
          }                                                            \
      }                                                                \
    while (0)
#else
#  define PSTAMP(i)
#endif

    // the constrained (Dirichlet) DoFs: dst = fused update of the identity row, i.e. with the cell operator's result
    // replaced by the source value (0 for the plain products); A x = x there (matrix_free_internal.h:247-255), P^-1 has
    // unit rows.  A DoF on an edge or corner lies on two or three planes: the plane of the lowest direction takes it (the
    // three-term update is not idempotent when dst is x_old).  One work item per DoF, grid-stride over the face workgroups.
    template <int MODE>
    __device__ __forceinline__ void
    constrained_dofs(const DevGrid &g, const TiledArgs &a, const long long first, const long long stride, const long long total)
    {
      const double *const opA = (MODE == OP_RESIDUAL) ? a.b : a.x;
      for (long long item = first; item < total; item += stride)
        {
          long long e = item;
          int       d = 0, side = 0;
          long long n1 = 1;
          bool      found = false;
#pragma unroll
          for (int f = 0; f < 6; ++f)
            if (!found)
              {
                const int fd = f / 2, fs = f % 2;
                if (fs ? g.dir_hi[fd] : g.dir_lo[fd])
                  {
                    const int       d1 = (fd + 1) % 3, d2 = (fd + 2) % 3;
                    const long long m1 = g.wrap[d1] ? g.nc[d1] * g.p : g.ext_hi[d1] - g.ext_lo[d1] + 1,
                                    m2 = g.wrap[d2] ? g.nc[d2] * g.p : g.ext_hi[d2] - g.ext_lo[d2] + 1;
                    if (e < m1 * m2)
                      {
                        d     = fd;
                        side  = fs;
                        n1    = m1;
                        found = true;
                      }
                    else
                      e -= m1 * m2;
                  }
              }
          if (!found)
            continue;
          const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
          int       c[3];
          c[d]  = side ? g.nc[d] * g.p : 0;
          c[d1] = (g.wrap[d1] ? 0 : g.ext_lo[d1]) + (int)(e % n1);
          c[d2] = (g.wrap[d2] ? 0 : g.ext_lo[d2]) + (int)(e / n1);
          bool earlier = false;
          for (int dd = 0; dd < d; ++dd)
            earlier = earlier || g.dirichlet(dd, c[dd]);
          if (earlier)
            continue;
          const long long o = g.offset(c[0], c[1], c[2]);
          const double    t = (MODE == OP_PLAIN || MODE == PC_PLAIN) ? 0. : a.src[o];
          const double    oa = ModeTraits<MODE>::need_a ? opA[o] : 0., ob = ModeTraits<MODE>::need_b ? a.x_old[o] : 0.;
          a.dst[o]           = fused_update<MODE>(a, t, oa, ob);
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // NW = waves per tile: 2 (one wave per SIMD, 256 + 256 registers) or 4 (two waves per SIMD, 128 + 128)
    template <int MODE, int NW>
    __global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
    pair_sweep_kernel(const DevGrid g, const DevKron kr, const TiledArgs a, const PairTiling tl)
    {
      extern __shared__ __attribute__((aligned(16))) double lds[];
      using MT = ModeTraits<MODE>;
      using AR = ARegs<MT::need_b, NW>;
      static_assert(NW == 2 || (NW == 4 && !MT::need_b), "four waves per tile: not for the three-term update");
      constexpr bool LDSFULL = !MT::need_b; // (see XYMatrices)
      constexpr bool FULLREGS = NW == 2;
      constexpr int  PPW     = 4 / NW;      // planes per wave and stage
      constexpr int  NZC     = z_columns(NW);
      constexpr int  NT      = 64 * NW;
      // workgroups b and b + 8 land on the same XCD: contiguous ranges of the work per XCD (L2 locality of neighbouring tiles)
      // Two-phase launches (TiledArgs::phase 1 / 2; the ghost exchange of the result runs beside phase 2): a.pair_phase_order
      // lists the workgroups at the block's rank interfaces first, padded with others to whole rounds (pair_phase_order);
      // phase 1 = its first a.pair_n_first entries and the Dirichlet planes, phase 2 the rest.
      // (the launcher has resolved the phase: a.pair_n_march workgroups march, their work indices are at a.pair_order +
      //  a.pair_order_offset)
      const int n_march = a.pair_n_march;
      if ((int)blockIdx.x >= n_march)
        {
          // (the last workgroups of the launch: the Dirichlet planes)
          constrained_dofs<MODE>(g, a, ((long long)blockIdx.x - n_march) * NT + threadIdx.x, (long long)tl.n_face * NT, tl.face_dofs);
          return;
        }
      int wi = (int)xcd_block(blockIdx.x, n_march);
      if (a.pair_order != nullptr) // expensive workgroups first (pair_order) / phase order; a scalar load
        wi = ((const __attribute__((address_space(4))) int *)a.pair_order)[wi + a.pair_order_offset];
      // (integer division runs on the vector unit even for wave-uniform values: back to scalar registers explicitly, or
      //  everything derived from the chunk index - the plane addresses the asm requests take as scalar operands - stays there)
      const int      n_tiles = tl.n_tiles();
      const int      ti = __builtin_amdgcn_readfirstlane(wi % n_tiles), zc = __builtin_amdgcn_readfirstlane(wi / n_tiles);
      const TileGeom tg = tile_geom(tl, ti);
      const int      X0 = __builtin_amdgcn_readfirstlane(tg.X0), Y0 = __builtin_amdgcn_readfirstlane(tg.Y0);
#ifndef ASM_PAIR_ABL
#  define ASM_PAIR_ABL 0 // diagnostic builds (wrong results): 1 no plane requests, 2 no diagonal reads, 4 no scratchpad traffic, 8 no arithmetic in the z columns
#endif
#ifdef ASM_PAIR_STAMP
#  ifndef ASM_PAIR_STAMP_TILE
#    define ASM_PAIR_STAMP_TILE 0 // 0: a tile in the middle of the block, 1: at the low x face, 2: at the low x and y faces
#  endif
      const bool stamp_on = ti == (ASM_PAIR_STAMP_TILE >= 1 ? 0 : tl.ntx / 2) + tl.ntx * (ASM_PAIR_STAMP_TILE >= 2 ? 0 : tl.nty / 2) && zc == tl.nzc / 2 &&
                            threadIdx.x < 64;
      unsigned long long stamp_t = __builtin_amdgcn_s_memtime(), stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

      const int lane = threadIdx.x & 63, cl = lane & ((1 << tg.lg_tx) - 1), rw = lane >> tg.lg_tx;
      const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 .. NW - 1, wave-uniform
      const int cx = X0 + cl, cy = Y0 + rw;
      Lane      c;
      int       clx = 0, cly = 0;
      bool      cell_xy, owner, ph_x, ph_y;
      {
        const bool comp_x = cell_computable(g, 0, cx), comp_y = cell_computable(g, 1, cy);
        ph_x              = cx == g.nc[0] && has_phantom(g, 0);
        ph_y              = cy == g.nc[1] && has_phantom(g, 1);
        cell_xy           = comp_x && comp_y;
        // a cell's DoFs (0..3)^2 belong to the tile that holds it outside its halo column / row; cells -1 are never owned
        const bool own_x = cx >= 0 && (comp_x || ph_x) && (cl >= 1 || tg.lo_x) && cx <= tg.x_max;
        const bool own_y = cy >= 0 && (comp_y || ph_y) && (rw >= 1 || tg.lo_y);
        c.tw             = 1 << tg.lg_tx;
        owner            = own_x && own_y;
        c.flags          = (cell_xy ? F_CELL : 0u) | (owner ? F_OWNER : 0u) | (ph_x ? F_PH_X : 0u) | (ph_y ? F_PH_Y : 0u) |
                  ((owner && g.dirichlet(0, P * cx)) ? F_XD0 : 0u) | ((owner && g.dirichlet(1, P * cy)) ? F_YD0 : 0u) | (cl >= 1 ? F_LEFT : 0u) |
                  (rw >= 1 ? F_BELOW : 0u);
        if (kr.mask_constrained && cell_xy)
          c.flags |= (g.dirichlet(0, P * cx) ? F_SM_X0 : 0u) | (g.dirichlet(0, P * cx + P) ? F_SM_X4 : 0u) | (g.dirichlet(1, P * cy) ? F_SM_Y0 : 0u) |
                     (g.dirichlet(1, P * cy + P) ? F_SM_Y4 : 0u);
        // source rows: the cell itself, or cell (0, 0) for lanes without a computed cell (any finite data: zero diagonal)
        const int sx = cell_xy ? cx : 0, sy = cell_xy ? cy : 0;
        const int x0 = g.store_index_near(0, P * sx);
        c.sd4        = (unsigned)(g.store_index_near(0, P * sx + P) - x0) * 8u;
        c.sstep      = (unsigned)g.stride[1] * 8u;
        c.srow0      = (unsigned)(g.store_index_near(1, P * sy) * (int)g.stride[1] + x0) * 8u;
        c.srow4      = (unsigned)(g.store_index_near(1, P * sy + P) * (int)g.stride[1] + x0) * 8u;
        // owned rows: rows 1..3 of a cell that only owns row 0 do not exist
        const int ox = owner ? cx : 0, oy = owner ? cy : 0;
        const int xo = g.store_index_near(0, P * ox);
        c.orow0      = (unsigned)(g.store_index_near(1, P * oy) * (int)g.stride[1] + xo) * 8u;
        c.ostep      = (owner && ph_y) ? 0u : c.sstep;
        if (cell_xy)
          {
            clx = kr.cls[0][cx + 1];
            cly = kr.cls[1][cy + 1];
          }
      }

      const lds_ptr Xl = (lds_ptr)lds + lane;
      const lds_ptr Dl = (lds_ptr)lds + L_X;
      const lds_ptr Zl = Dl + L_D;
      const lds_ptr Ml = Zl + L_ZERO;
      for (int e = threadIdx.x; e < L_M; e += NT)
        Ml[e] = kr.mat[e];
      for (int e = threadIdx.x; e < L_ZERO; e += NT)
        Zl[e] = 0.;
      XYMatrices mxy;
      mxy.mx = Ml + (0 * KRON_MAX_CLS + clx) * KRON_MAT_STRIDE;
      mxy.my = Ml + (1 * KRON_MAX_CLS + cly) * KRON_MAT_STRIDE;
      {
        const bool nsx = cell_xy && kr.sym[0 * KRON_MAX_CLS + clx] == 0, nsy = cell_xy && kr.sym[1 * KRON_MAX_CLS + cly] == 0;
        c.flags |= (nsx ? F_NSX : 0u) | (nsy ? F_NSY : 0u);
        mxy.any_nsx = __any(nsx);
        mxy.any_nsy = __any(nsy);
        mxy.clx     = clx;
        mxy.cly     = cly;
        mxy.gx      = kr.mat + 0 * KRON_MAX_CLS * KRON_MAT_STRIDE;
        mxy.gy      = kr.mat + 1 * KRON_MAX_CLS * KRON_MAT_STRIDE;
      }
      // this lane's diagonal: column e of this wave at Dz + 5 e (cells that are not computed: zeros)
      const int      col0 = role == 0 ? 0 : NZC + (role - 1) * (NZC - 1);
      const lds_cptr Dz   = cell_xy ? Dl + (cly * KRON_MAX_CLS + clx) * KRON_DIAG_STRIDE + col0 * N1 : Zl;

      // scratchpad addresses of this lane: forward planes of this wave, its z columns, its backward planes
      const lds_ptr X_fwd = Xl + (PPW * role) * NE * 64;
      const lds_ptr X_z   = Xl + col0 * 64;
      const lds_ptr X_bwd = Xl + (PPW * role) * NE * 64;
      const int     kf = 1 + PPW * role, kb = PPW * role; // first forward / backward plane of this wave

      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;

      const double *const opA = (MODE == OP_RESIDUAL) ? a.b : a.x;

      // carried by the z sweeps, per column of this wave: plane 4 of the layer below in transformed space (t0) and the
      // layer below's contribution to this layer's plane 0 (carry); NW = 4: in accumulation registers
      constexpr bool CARRY_AREGS = NW == 4;
      double         t0[CARRY_AREGS ? 1 : NZC], carry[CARRY_AREGS ? 1 : NZC];
      const auto     t0_get = [&](auto ec) -> double {
        constexpr int e = decltype(ec)::value;
        if constexpr (CARRY_AREGS)
          return areg_double<AR::T0 + 2 * e>();
        else
          return t0[e];
      };
      const auto t0_set = [&](auto ec, const double x) {
        constexpr int e = decltype(ec)::value;
        if constexpr (CARRY_AREGS)
          areg_set<AR::T0 + 2 * e>(x);
        else
          t0[e] = x;
      };
      const auto carry_get = [&](auto ec) -> double {
        constexpr int e = decltype(ec)::value;
        if constexpr (CARRY_AREGS)
          return areg_double<AR::CY + 2 * e>();
        else
          return carry[e];
      };
      const auto carry_set = [&](auto ec, const double x) {
        constexpr int e = decltype(ec)::value;
        if constexpr (CARRY_AREGS)
          areg_set<AR::CY + 2 * e>(x);
        else
          carry[e] = x;
      };

      areg_reserve<NW>();
      static_for<NZC>([&](auto ec) { carry_set(ec, 0.); });
      __syncthreads(); // tables in LDS
      int clz_cached = -1;
      // the z classes: this layer's was loaded a layer ago; which of them have an even-odd form
      int      clz_next = uniform_load(kr.cls[2], z_first + 1);
      unsigned sym_z    = 0u;
#pragma unroll
      for (int q = 0; q < KRON_MAX_CLS; ++q)
        sym_z |= (uniform_load(kr.sym, 2 * KRON_MAX_CLS + q) != 0 ? 1u : 0u) << q;
      // ---- prologue: plane 0 of the first layer (wave 0 transforms it, both pick their columns), first requests
      {
        const long long z0 = plane_offset(g, P * z_first);
        plane_request<AR::P0>(a.src + z0, c);
        if (PPW == 2)
          {
            plane_request<AR::S0>(a.src + plane_offset(g, P * z_first + kf), c);
            plane_request<AR::S1>(a.src + plane_offset(g, P * z_first + kf + 1), c);
          }
        vm_wait<0>();
        double v[NE];
        plane_unpack<AR::P0>(v);
        // (one plane per wave: the prologue's plane came through the registers of the first layer's plane)
        if (PPW == 1)
          plane_request<AR::S0>(a.src + plane_offset(g, P * z_first + kf), c);
        if (kr.mask_constrained)
          mask_source(v, c, g.dirichlet(2, P * z_first));
        // (the first operands go into registers the prologue's plane occupied)
        if (MT::need_a && z_first >= Zb)
          {
            operand_request<AR::XA0>(opA + plane_offset(g, P * z_first + kb), c);
            if (PPW == 2)
              operand_request<AR::XA1>(opA + plane_offset(g, P * z_first + kb + 1), c);
          }
        if (MT::need_b && z_first >= Zb)
          {
            operand_request<AR::XB0>(a.x_old + plane_offset(g, P * z_first + kb), c);
            if (PPW == 2)
              operand_request<AR::XB1>(a.x_old + plane_offset(g, P * z_first + kb + 1), c);
          }
        plane2d<true, LDSFULL, FULLREGS>(v, mxy, c);
        if (role == 0)
          {
#pragma unroll
            for (int e = 0; e < NE; ++e)
              Xl[e * 64] = v[e];
          }
        __syncthreads();
        static_for<NZC>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          t0_set(ec, (e < NZC - 1 || role == 0) ? X_z[e * 64] : 0.);
        });
        __syncthreads();
      }

      for (int cz = z_first; cz < Ze; ++cz)
        {
          const bool emit = cz >= Zb, more = cz + 1 < Ze;
          c.refresh();
          PSTAMP(0);
          // the diagonals of this layer's z class
          const int clz = clz_next;
          clz_next      = uniform_load(kr.cls[2], min(cz + 2, g.nc[2] + 1));
          if (clz != clz_cached)
            {
              // (the scratchpad barriers below order these writes against the reads of the layer before)
              // (rare - first layer, boundary layers: a rolled loop over an index the compiler cannot trace, or it keeps the
              //  unrolled copies' addresses in registers over the whole march)
              const double *src = kr.diag + (long long)clz * L_D;
              int           e0  = threadIdx.x;
              asm volatile("" : "+v"(e0));
#pragma unroll 1
              for (int e = e0; e < L_D; e += NT)
                Dl[e] = src[e];
              clz_cached = clz;
            }
          PSTAMP(6);
          // ---- forward x-y transforms of this wave's two planes -> scratchpad
          {
            if (cz == z_first || cz == Zb || MT::n_operand_requests == 0)
              vm_wait<0>(); // first layers of a chunk, or nothing but the stores of the layer below is younger
            else
              vm_wait<PPW * MT::n_operand_requests>(); // younger than the plane requests: the operand requests of this layer
            PSTAMP(7);
            {
              double v[NE];
              plane_unpack<AR::S0>(v);
              if (kr.mask_constrained)
                mask_source(v, c, g.dirichlet(2, P * cz + kf)); // (only plane 4 of a layer can be a Dirichlet plane)
              plane2d<true, LDSFULL, FULLREGS>(v, mxy, c);
#pragma unroll
              for (int e = 0; e < NE; ++e)
                X_fwd[(0 * NE + e) * 64] = v[e];
            }
            if (PPW == 2)
              {
                double v[NE];
                plane_unpack<AR::S1>(v);
                if (kr.mask_constrained)
                  mask_source(v, c, g.dirichlet(2, P * cz + kf + 1));
                plane2d<true, LDSFULL, FULLREGS>(v, mxy, c);
#pragma unroll
                for (int e = 0; e < NE; ++e)
                  X_fwd[(1 * NE + e) * 64] = v[e];
              }
            PSTAMP(8);
          }
          // the next layer's planes are requested a whole layer ahead, one row per z column below
          const double *const next0 = a.src + plane_offset(g, P * (cz + (more ? 1 : 0)) + kf);
          const double *const next1 = a.src + plane_offset(g, P * (cz + (more ? 1 : 0)) + kf + 1);
          const lds_cptr      mz    = Ml + (2 * KRON_MAX_CLS + clz) * KRON_MAT_STRIDE;
          const bool          eo_z  = ((sym_z >> clz) & 1u) != 0u;
          PSTAMP(1);
          __syncthreads();
          PSTAMP(2);
          // ---- z sweeps of this wave's columns and the diagonal, in place in the scratchpad; reads run one column ahead
          {
            double n1 = X_z[(0 * NE) * 64], n2 = X_z[(1 * NE) * 64], n3 = X_z[(2 * NE) * 64], n4 = X_z[(3 * NE) * 64];
            double nd[N1];
#pragma unroll
            for (int cc = 0; cc < N1; ++cc)
              nd[cc] = Dz[cc];
            const auto columns = [&](const auto &qf, const auto &qb) {
              static_for<NZC>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if (e < NZC - 1 || role == 0)
                  {
                    if (more && !(ASM_PAIR_ABL & 1))
                      plane_request_step<AR::S0, AR::S1, PPW>(e, next0, next1, c);
                    double       u0 = t0_get(ec), u1 = n1, u2 = n2, u3 = n3, u4 = n4;
                    const double d0 = nd[0], d1 = nd[1], d2 = nd[2], d3 = nd[3], d4 = nd[4];
                    if (e + 1 < NZC && (e + 1 < NZC - 1 || role == 0))
                      {
                        if (!(ASM_PAIR_ABL & 4))
                          {
                            n1 = X_z[(0 * NE + e + 1) * 64];
                            n2 = X_z[(1 * NE + e + 1) * 64];
                            n3 = X_z[(2 * NE + e + 1) * 64];
                            n4 = X_z[(3 * NE + e + 1) * 64];
                          }
                        if (!(ASM_PAIR_ABL & 2))
                          {
#pragma unroll
                            for (int cc = 0; cc < N1; ++cc)
                              nd[cc] = Dz[(e + 1) * N1 + cc];
                          }
                      }
                    t0_set(ec, u4);
                    if (!(ASM_PAIR_ABL & 8))
                      {
                        line5<true>(u0, u1, u2, u3, u4, qf);
                        u0 *= d0;
                        u1 *= d1;
                        u2 *= d2;
                        u3 *= d3;
                        u4 *= d4;
                        line5<false>(u0, u1, u2, u3, u4, qb);
                      }
                    if (emit && !(ASM_PAIR_ABL & 4))
                      {
                        X_z[(0 * NE + e) * 64] = u0 + carry_get(ec);
                        X_z[(1 * NE + e) * 64] = u1;
                        X_z[(2 * NE + e) * 64] = u2;
                        X_z[(3 * NE + e) * 64] = u3;
                      }
                    carry_set(ec, u4);
                  }
              });
            };
            if (eo_z)
              columns(EoCoef(mz + 60), EoCoef(mz + 76));
            else if (LDSFULL)
              columns(FullCoef(mz), FullCoef(mz + 30)); // first / last layer of a block with Dirichlet or Neumann faces
            else
              {
                const double *gz = kr.mat + (long long)(2 * KRON_MAX_CLS + clz) * KRON_MAT_STRIDE;
                columns(FullCoefS(gz), FullCoefS(gz + 30));
              }
          }
          PSTAMP(3);
          __syncthreads();
          PSTAMP(4);
          if (emit)
            {
              // ---- backward x-y transforms of this wave's two output planes, merge, update, store
              const long long zo0 = plane_offset(g, P * cz + kb), zo1 = zo0 + g.stride[2]; // (planes 0..3 of a layer are consecutive)
              const bool      con0 = g.dirichlet(2, P * cz + kb);                           // (only plane 0 of a layer can be one)
              double          w[NE];
#pragma unroll
              for (int e = 0; e < NE; ++e)
                w[e] = X_bwd[(0 * NE + e) * 64];
              PSTAMP(9);
              // (no barrier before the next layer's planes go to the scratchpad: a wave writes the slots it has just read
              //  itself - output plane q and new plane q + 1 share slot q -, and a wave's LDS operations execute in order)
              if (cz == Zb || !more)
                vm_wait<0>();
              else
                vm_wait<PPW * 15>(); // younger than the operand requests: the plane requests of this layer
              PSTAMP(10);
              plane2d<false, LDSFULL, FULLREGS>(w, mxy, c);
              emit_plane<MODE, AR::XA0, AR::XB0>(a, c, a.dst + zo0, w, con0);
              // (the next layer's operands of this plane: its registers are free)
              if (more)
                {
                  if (MT::need_a)
                    operand_request<AR::XA0>(opA + plane_offset(g, P * (cz + 1) + kb), c);
                  if (MT::need_b)
                    operand_request<AR::XB0>(a.x_old + plane_offset(g, P * (cz + 1) + kb), c);
                }
              PSTAMP(11);
              if (PPW == 2)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    w[e] = X_bwd[(1 * NE + e) * 64];
                  PSTAMP(12);
                  PSTAMP(13);
                  plane2d<false, LDSFULL, FULLREGS>(w, mxy, c);
                  emit_plane<MODE, AR::XA1, AR::XB1>(a, c, a.dst + zo1, w, false);
                  if (more)
                    {
                      if (MT::need_a)
                        operand_request<AR::XA1>(opA + plane_offset(g, P * (cz + 1) + kb + 1), c);
                      if (MT::need_b)
                        operand_request<AR::XB1>(a.x_old + plane_offset(g, P * (cz + 1) + kb + 1), c);
                    }
                }
              PSTAMP(14);
            }
          else
            {
              if (more)
                {
                  // halo layer: the update operands of the first owned layer
                  if (MT::need_a)
                    {
                      operand_request<AR::XA0>(opA + plane_offset(g, P * (cz + 1) + kb), c);
                      if (PPW == 2)
                        operand_request<AR::XA1>(opA + plane_offset(g, P * (cz + 1) + kb + 1), c);
                    }
                  if (MT::need_b)
                    {
                      operand_request<AR::XB0>(a.x_old + plane_offset(g, P * (cz + 1) + kb), c);
                      if (PPW == 2)
                        operand_request<AR::XB1>(a.x_old + plane_offset(g, P * (cz + 1) + kb + 1), c);
                    }
                }
            }
        }
#ifdef ASM_PAIR_STAMP
      PSTAMP(5);
      if (stamp_on && lane == 0)
        {
          for (int i = 0; i < 16; ++i)
            atomicAdd(&g_pair_stamps[MODE][i], stamp_acc[i]);
          atomicAdd(&g_pair_stamps[MODE][16], (unsigned long long)(Ze - z_first));
        }
#endif
      if (Ze == g.nc[2] && has_phantom(g, 2))
        {
          // top plane of the block: what the last layer left in the carries of both waves, transformed back by wave 0
          __syncthreads();
          static_for<NZC>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            if (e < NZC - 1 || role == 0)
              X_z[e * 64] = carry_get(ec);
          });
          __syncthreads();
          if (role == 0 && !g.dirichlet(2, P * g.nc[2]))
            {
              double v[NE];
#pragma unroll
              for (int e = 0; e < NE; ++e)
                v[e] = Xl[e * 64];
              const long long zo = plane_offset(g, P * g.nc[2]);
              if (MT::need_a)
                operand_request<AR::XA0>(opA + zo, c);
              if (MT::need_b)
                operand_request<AR::XB0>(a.x_old + zo, c);
              vm_wait<0>();
              plane2d<false, LDSFULL, FULLREGS>(v, mxy, c);
              emit_plane<MODE, AR::XA0, AR::XB0>(a, c, a.dst + zo, v, false);
            }
        }
    }

    // ---- host side ------------------------------------------------------------------------------------------------
    // relative cost of a workgroup's layer: tiles with cells at a physical boundary take the sweeps of that direction with
    // full matrices (stage timers: profiles/r04_pair_stamps.txt)
    double
    tile_cost(const DevGrid &g, const PairTiling &tl, const int t)
    {
      const TileGeom q  = tile_geom(tl, t);
      const int      tw = 1 << q.lg_tx, th = 64 >> q.lg_tx;
      const auto at_boundary = [&](const int d, const int c0, const int width) {
        // cells c0 .. c0 + width - 1 of direction d; the first / last cell of a block at a physical boundary
        const bool lo = !g.wrap[d] && g.ext_lo[d] == 0, hi = !g.wrap[d] && g.own_hi[d] == g.nc[d] * P;
        return (lo && c0 <= 0) || (hi && c0 <= g.nc[d] - 1 && c0 + width > g.nc[d] - 1);
      };
      // (stage timers: +17 % per layer in a tile at a face, about twice that at an edge)
      return 1.0 + (at_boundary(0, q.X0, tw) ? 0.17 : 0.) + (at_boundary(1, q.Y0, th) ? 0.17 : 0.);
    }
    int
    chunk_layers(const DevGrid &g, const PairTiling &tl, const int zc)
    {
      const int  Zb = zc * tl.zlen, Ze = std::min(Zb + tl.zlen, g.nc[2]);
      const bool halo = Zb > 0 || g.wrap[2] || g.ext_lo[2] < 0;
      return Ze - Zb + (halo ? 1 : 0);
    }
    // launch position -> work index (tile x fastest, tile y, z chunk): every XCD (workgroups b, b + 8, ...) keeps a
    // contiguous range of the work - neighbouring tiles share cache lines in its L2 - but takes its expensive workgroups
    // first; returns the time the last workgroup ends (in layer times) when `slots` workgroups run at a time
    double
    schedule(const DevGrid &g, const PairTiling &tl, const long long slots, std::vector<int> *order)
    {
      const int           nt = tl.n_tiles(), n = nt * tl.nzc;
      std::vector<double> cost(n);
      for (int w = 0; w < n; ++w)
        cost[w] = tile_cost(g, tl, w % nt) * (chunk_layers(g, tl, w / nt) + 0.5);
      std::vector<int> pos(n);
      for (int xcd = 0; xcd < 8; ++xcd)
        {
          const int begin = xcd * (n >> 3) + std::min(xcd, n & 7), count = (n >> 3) + (xcd < (n & 7) ? 1 : 0);
          std::vector<int> w(count);
          for (int i = 0; i < count; ++i)
            w[i] = begin + i;
          std::stable_sort(w.begin(), w.end(), [&](const int x, const int y) { return cost[x] > cost[y]; });
          for (int i = 0; i < count; ++i)
            pos[begin + i] = w[i];
        }
      if (order != nullptr)
        *order = pos;
      // list scheduling in dispatch order: block b takes launch position xcd_block(b)
      std::vector<double> free_at((size_t)std::max<long long>(1, std::min<long long>(slots, n)), 0.);
      std::make_heap(free_at.begin(), free_at.end(), std::greater<double>());
      double end = 0.;
      for (int bb = 0; bb < n; ++bb)
        {
          std::pop_heap(free_at.begin(), free_at.end(), std::greater<double>());
          const double t = free_at.back() + cost[pos[(int)xcd_block(bb, n)]];
          free_at.back() = t;
          std::push_heap(free_at.begin(), free_at.end(), std::greater<double>());
          end = std::max(end, t);
        }
      return end;
    }

    PairTiling
    make_pair_tiling(const DevGrid &g, const int n_cu_in, const int zchunks = 0, std::vector<int> *ranking = nullptr)
    {
      // (a pure function of the block and the device: computed once per shape)
      struct Key
      {
        int v[20];
        bool operator<(const Key &o) const { return std::lexicographical_compare(v, v + 20, o.v, o.v + 20); }
      };
      static std::mutex                mtx;
      static std::map<Key, PairTiling> cache;
      Key                              key;
      for (int d = 0; d < 3; ++d)
        {
          key.v[d]      = g.nc[d];
          key.v[3 + d]  = g.wrap[d];
          key.v[6 + d]  = g.ext_lo[d];
          key.v[9 + d]  = g.own_hi[d];
          key.v[12 + d] = g.ext_hi[d];
          key.v[17 + d] = g.dir_lo[d] * 2 + g.dir_hi[d];
        }
      const char *env = std::getenv("ASM_PAIR_ZCHUNKS");
      key.v[15]       = n_cu_in;
      key.v[16]       = env != nullptr ? std::atoi(env) : zchunks;
      if (ranking == nullptr)
        {
          std::lock_guard<std::mutex> lock(mtx);
          const auto                  it = cache.find(key);
          if (it != cache.end())
            return it->second;
        }
      PairTiling tl;
      const int  n_cu  = n_cu_in > 0 ? n_cu_in : 256;
      const auto first = [&](const int d) { return (g.wrap[d] || g.ext_lo[d] < 0) ? -1 : 0; };
      const auto tiles = [&](const int d, const int first_cell, const int owned) {
        const int last = g.nc[d] - 1 + (has_phantom(g, d) ? 1 : 0); // last cell that owns DoFs
        // tile t owns the cells first + t owned + 1 .. first + (t + 1) owned, tile 0 also its first column if that is cell 0
        int n = 1;
        while (first_cell + n * owned < last)
          ++n;
        return n;
      };
      tl.x_first = first(0);
      tl.y_first = first(1);
      tl.ntx     = tiles(0, tl.x_first, OX);
      tl.nty     = tiles(1, tl.y_first, OY);
      tl.n_rs = tl.n_ts = 0;
      tl.xs0 = tl.ys0 = 0;
      {
        // strips instead of a last tile column / row that would own 1..3 cell columns / rows: whichever of the four
        // combinations has the fewest tiles (ASM_PAIR_NO_STRIPS: experiments)
        const int  last[2] = {g.nc[0] - 1 + (has_phantom(g, 0) ? 1 : 0), g.nc[1] - 1 + (has_phantom(g, 1) ? 1 : 0)};
        const int  full[2] = {tl.ntx, tl.nty}, frst[2] = {tl.x_first, tl.y_first}, own[2] = {OX, OY};
        bool       can[2];
        for (int d = 0; d < 2; ++d)
          {
            const int left = last[d] - (frst[d] + (full[d] - 1) * own[d]); // cells the last tile column / row would own
            can[d]         = full[d] >= 2 && left >= 1 && left <= 3 && std::getenv("ASM_PAIR_NO_STRIPS") == nullptr;
          }
        const auto strips = [&](const int first_cell, const int upto) { // tiles of 15 owned cells that own everything up to `upto`
          int n = 1;
          while (first_cell + n * 15 < upto)
            ++n;
          return n;
        };
        long long best_n = (long long)tl.ntx * tl.nty;
        for (int sx = 0; sx <= (can[0] ? 1 : 0); ++sx)
          for (int sy = 0; sy <= (can[1] ? 1 : 0); ++sy)
            {
              if (sx + sy == 0)
                continue;
              const int ntx = full[0] - sx, nty = full[1] - sy;
              const int xs0 = tl.x_first + ntx * OX, ys0 = tl.y_first + nty * OY;
              const int n_rs = sx ? strips(tl.y_first, last[1]) : 0;              // right strips: all rows
              const int n_ts = sy ? strips(tl.x_first, sx ? xs0 : last[0]) : 0; // upper strips: the main tiles' columns
              const long long n = (long long)ntx * nty + n_rs + n_ts;
              if (n < best_n)
                {
                  best_n  = n;
                  tl.ntx  = ntx;
                  tl.nty  = nty;
                  tl.n_rs = n_rs;
                  tl.n_ts = n_ts;
                  tl.xs0  = xs0;
                  tl.ys0  = ys0;
                }
            }
      }
      // z chunks: two workgroups per compute unit run at a time; a chunk costs its layers plus the halo layer below it,
      // tiles at physical boundaries cost more per layer: the chunk count whose simulated schedule ends first
      const long long slots = 2ll * n_cu;
      int             best = 1;
      if (key.v[16] > 0)
        best = std::max(1, std::min(key.v[16], g.nc[2]));
      else
        {
          double                              best_end = 1e300;
          int                                 last_chunks = 0;
          std::vector<std::pair<double, int>> scored;
          for (int nzc = 1; nzc <= g.nc[2]; ++nzc)
            {
              tl.zlen = (g.nc[2] + nzc - 1) / nzc;
              tl.nzc  = (g.nc[2] + tl.zlen - 1) / tl.zlen;
              if (tl.nzc == last_chunks || (long long)tl.n_tiles() * tl.nzc > 16384)
                continue;
              last_chunks      = tl.nzc;
              const double end = schedule(g, tl, slots, nullptr);
              scored.emplace_back(end, tl.nzc);
              if (end < best_end * (1. - 1e-9))
                {
                  best_end = end;
                  best     = tl.nzc;
                }
            }
          if (ranking != nullptr)
            {
              // (the model ranks chunk counts whose schedules end within a few per cent of each other: the context
              //  measures the first few, context.hip: tune_pair)
              std::sort(scored.begin(), scored.end());
              ranking->clear();
              for (const auto &sc : scored)
                ranking->push_back(sc.second);
            }
        }
      tl.zlen = (g.nc[2] + best - 1) / best;
      tl.nzc  = (g.nc[2] + tl.zlen - 1) / tl.zlen;
      // workgroups for the Dirichlet planes: eight DoFs per thread, at most one round of compute units' worth
      tl.face_dofs = 0;
      for (int d = 0; d < 3; ++d)
        {
          const int       d1 = (d + 1) % 3, d2 = (d + 2) % 3;
          const long long m1 = g.wrap[d1] ? g.nc[d1] * g.p : g.ext_hi[d1] - g.ext_lo[d1] + 1, m2 = g.wrap[d2] ? g.nc[d2] * g.p : g.ext_hi[d2] - g.ext_lo[d2] + 1;
          tl.face_dofs += (long long)((g.dir_lo[d] ? 1 : 0) + (g.dir_hi[d] ? 1 : 0)) * m1 * m2;
        }
      tl.n_face = (int)std::min<long long>((tl.face_dofs + 128 * 8 - 1) / (128 * 8), 2ll * n_cu);
      std::lock_guard<std::mutex> lock(mtx);
      cache[key] = tl;
      return tl;
    }

    template <int MODE, int NW>
    bool
    launch_pair(const DevGrid &g, const DevKron &k, const TiledArgs &a, const PairTiling &tl, hipStream_t st)
    {
      static std::atomic<unsigned long long> done{0};
      const unsigned long long               bit = 1ull << (a.device & 63);
      if (!(done.load(std::memory_order_acquire) & bit))
        {
          if (hipFuncSetAttribute((const void *)pair_sweep_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(L_TOTAL * sizeof(double))) != hipSuccess)
            {
              (void)hipGetLastError();
              return false;
            }
          done.fetch_or(bit, std::memory_order_release);
        }
      // the phase, resolved for the kernel: how many workgroups march and where their work indices are
      TiledArgs b   = a;
      const int n_all = tl.n_tiles() * tl.nzc;
      b.pair_n_march  = a.phase == 0 ? n_all : (a.phase == 1 ? a.pair_n_first : n_all - a.pair_n_first);
      if (a.phase != 0)
        {
          b.pair_order        = a.pair_phase_order;
          b.pair_order_offset = a.phase == 2 ? a.pair_n_first : 0;
        }
      const int grid = b.pair_n_march + (a.phase != 2 ? tl.n_face : 0); // (the Dirichlet planes: with phase 1)
      if (grid > 0)
        hipLaunchKernelGGL((pair_sweep_kernel<MODE, NW>), dim3(grid), dim3(64 * NW), L_TOTAL * sizeof(double), st, g, k, b, tl);
      return true;
    }
    // four waves per tile unless the mode or the caller (TiledArgs::pair_waves: the context's measurement or ASM_PAIR_WAVES
    // when it was created) says otherwise
    template <int MODE>
    bool
    launch_pair(const DevGrid &g, const DevKron &k, const TiledArgs &a, const PairTiling &tl, hipStream_t st)
    {
      if constexpr (!ModeTraits<MODE>::need_b)
        if (a.pair_waves != 2)
          return launch_pair<MODE, 4>(g, k, a, tl, st);
      return launch_pair<MODE, 2>(g, k, a, tl, st);
    }
  } // namespace

#ifdef ASM_PAIR_STAMP
  // stage cycles of the diagnostic build (tools/pair_stamps.py); not part of the C ABI
  extern "C" void
  asm_debug_pair_stamps(unsigned long long out[6 * 24], int reset)
  {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), 6 * 24 * sizeof(unsigned long long));
    if (reset)
      {
        unsigned long long z[6 * 24] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stamps), z, sizeof(z));
      }
  }
#endif

  void
  pair_chunk_candidates(const DevGrid &g, const int n_cu, std::vector<int> &zchunks)
  {
    make_pair_tiling(g, n_cu, 0, &zchunks);
  }

  int
  pair_order(const DevGrid &g, const int n_cu, std::vector<int> &order, const int zchunks)
  {
    const PairTiling tl = make_pair_tiling(g, n_cu, zchunks);
    schedule(g, tl, 2ll * (n_cu > 0 ? n_cu : 256), &order);
    return (int)order.size();
  }

  // work indices of a two-phase launch: the workgroups with the first / last tile or chunk of a direction whose face is a
  // rank interface (they produce what the neighbours ghost), then - from position n_first = their number rounded up to whole
  // rounds of workgroup slots - the others; returns n_first
  int
  pair_phase_order(const DevGrid &g, const int n_cu, const bool iface[3][2], std::vector<int> &order, const int zchunks)
  {
    const PairTiling tl = make_pair_tiling(g, n_cu, zchunks);
    const int        nt = tl.n_tiles(), n = nt * tl.nzc;
    std::vector<int> inner;
    order.clear();
    for (int w = 0; w < n; ++w)
      {
        const TileGeom q   = tile_geom(tl, w % nt);
        const int      zc  = w / nt, c0[2] = {q.X0, q.Y0}, wd[2] = {1 << q.lg_tx, 64 >> q.lg_tx};
        bool           bd  = (iface[2][0] && zc == 0) || (iface[2][1] && zc == tl.nzc - 1);
        for (int d = 0; d < 2; ++d) // the tile holds the block's first / last cell of the direction
          bd = bd || (iface[d][0] && c0[d] <= 0) || (iface[d][1] && c0[d] + wd[d] > g.nc[d] - 1);
        (bd ? order : inner).push_back(w);
      }
    const long long slots   = 2ll * (n_cu > 0 ? n_cu : 256);
    const int       n_first = order.empty() ? 0 : (int)std::min<long long>(n, ((long long)order.size() + slots - 1) / slots * slots);
    order.insert(order.end(), inner.begin(), inner.end());
    return n_first;
  }

  void
  pair_plan(const DevGrid &g, const int n_cu, int out[6], const int zchunks)
  {
    const PairTiling tl = make_pair_tiling(g, n_cu, zchunks);
    out[0]              = tl.ntx;
    out[1]              = tl.nty;
    out[2]              = tl.nzc;
    out[3]              = tl.zlen;
    out[4]              = tl.x_first;
    out[5]              = tl.y_first;
  }

  // every tile of the tiling: {first cell x, y, cells in x, in y, owns its first column (1) / row (2), last column it may own}
  void
  pair_tiles(const DevGrid &g, const int n_cu, std::vector<std::array<int, 6>> &tiles, const int zchunks)
  {
    const PairTiling tl = make_pair_tiling(g, n_cu, zchunks);
    tiles.clear();
    for (int t = 0; t < tl.n_tiles(); ++t)
      {
        const TileGeom q = tile_geom(tl, t);
        tiles.push_back({q.X0, q.Y0, 1 << q.lg_tx, 64 >> q.lg_tx, (q.lo_x ? 1 : 0) | (q.lo_y ? 2 : 0), q.x_max});
      }
  }

  bool
  pair_sweep(const DevGrid &g, const DevKron &k, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || !k.valid || (a.phase != 0 && a.pair_phase_order == nullptr))
      return false;
    // 32-bit byte offsets inside a DoF plane
    if (g.stride[2] * 8 >= (1ll << 31))
      return false;
    const PairTiling tl = make_pair_tiling(g, a.n_cu, a.pair_zchunks);
    switch (a.mode)
      {
        case OP_PLAIN: return launch_pair<OP_PLAIN>(g, k, a, tl, st);
        case OP_RESIDUAL: return launch_pair<OP_RESIDUAL>(g, k, a, tl, st);
        case PC_PLAIN: return launch_pair<PC_PLAIN>(g, k, a, tl, st);
        case PC_FIRST: return launch_pair<PC_FIRST>(g, k, a, tl, st);
        case PC_STEP1: return launch_pair<PC_STEP1>(g, k, a, tl, st);
        case PC_UPDATE: return launch_pair<PC_UPDATE>(g, k, a, tl, st);
        default: return false;
      }
  }
} // namespace asmamd
