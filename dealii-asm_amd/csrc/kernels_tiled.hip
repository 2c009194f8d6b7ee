// Tiled owner-computes sweeps for degree 4, cell patches with n_overlap = 1 (gfx950).
//
// One workgroup (4 waves, one per SIMD) owns a tile of 15 x 15 cells' worth of DoFs
// in x-y and marches along z.  It computes 16 x 16 cells per layer (one halo row /
// column on the low sides), ONE CELL PER LANE, the whole 5x5x5 patch tile living in
// that lane's registers, so the 1-D eigenvector sweeps (TensorProductMatrix-
// SymmetricSum::apply_inverse as called at include/matrix_free.h:1046-1052) are pure
// register FMA chains with no LDS traffic.  LDS is used only to
//   * stage one (65 x 65) DoF plane of src at a time, read coalesced from HBM and
//     scattered to the cells that contain each DoF (the gather of
//     include/vector_access_reduced.h:267-405 without any index list), and
//   * exchange the cells' output planes, which the DoF owners sum in a fixed order
//     (the scatter-add of :407-548 without atomics, bitwise reproducible).
// Every owned DoF is written exactly once, by one thread that holds its final
// value, so the Chebyshev vector updates that the reference passes as pre/post
// lambdas (include/matrix_free.h:960-986, PreconditionChebyshev) are applied in the
// store epilogue and dst needs no zeroing, no compress(add) and no atomics.
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
    constexpr int TC  = 16;       // computed cells per tile and direction
    constexpr int TO  = TC - 1;   // owned cells per tile and direction
    constexpr int NT  = TC * TC;  // threads = cells per layer
    constexpr int SP  = NT + 4;   // staging pitch (bank spread for the 4-lane cell groups)
    constexpr int NR  = 17;       // rows handled by one loader thread (rows wave + 4 r)
    constexpr int EIG = NE + N1;  // eigen data per 1-D instance

    struct Tiling
    {
      int ntx, nty, nzc; // tiles in x, y; chunks in z
      int zlen;          // owned cell layers per chunk
    };

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

    // is tile-relative DoF coordinate Ir (logical I) written by this tile?
    __device__ __forceinline__ bool
    tile_owns(const DevGrid &g, const int d, const int Ir, const int I)
    {
      if (I < 0 || I > g.own_hi[d] || Ir < P)
        return false;
      return Ir < 4 * TC || I == g.nc[d] * P;
    }

    template <bool FWD>
    __device__ __forceinline__ void
    line5(double &u0, double &u1, double &u2, double &u3, double &u4, const double (&q)[NE])
    {
      // FWD: out[a] = sum_i q[i][a] in[i];  else out[i] = sum_a q[i][a] in[a]
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

    // x and y sweeps on one 5x5 plane v[j*5+i]
    template <bool FWD>
    __device__ __forceinline__ void
    plane_x(double (&v)[NE], const double (&q)[NE])
    {
#pragma unroll
      for (int j = 0; j < N1; ++j)
        line5<FWD>(v[j * 5 + 0], v[j * 5 + 1], v[j * 5 + 2], v[j * 5 + 3], v[j * 5 + 4], q);
    }
    template <bool FWD>
    __device__ __forceinline__ void
    plane_y(double (&v)[NE], const double (&q)[NE])
    {
#pragma unroll
      for (int i = 0; i < N1; ++i)
        line5<FWD>(v[0 * 5 + i], v[1 * 5 + i], v[2 * 5 + i], v[3 * 5 + i], v[4 * 5 + i], q);
    }

    __device__ __forceinline__ void
    load_q(double (&q)[NE], const double *__restrict__ src)
    {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        q[e] = src[e];
    }

    // -------------------------------------------------------------------------
    // loader / assembler roles: thread (lane, wave) handles tile-relative DoF
    // column Ir = lane and rows Jr = wave + 4 r (r = 0..16); lanes <= 16 of every
    // wave additionally handle column Ir = 64, row Jr = wave + 4 lane.
    // Everything that does not depend on the plane is computed once per kernel.
    // -------------------------------------------------------------------------
    constexpr int PITCH = 4 * TC + 2; // LDS pitch of the raw input plane (doubles)

    struct PlaneRole
    {
      int       lane, wave;
      int       J0, ny, wrap_y, lo_y; // to rebuild the y offset of row r
      long long stride_y;
      unsigned  row_exists;    // bit r: row exists in the stored box
      unsigned  row_owned;     // bit r: row is written by this tile
      unsigned  row_dirichlet; // bit r: row is a Dirichlet plane
      long long col_off;       // x part of the offset of column Ir = lane
      bool      col_exists, col_owned, col_dirichlet;
      // column 64 element
      long long x_off;
      bool      x_exists, x_owned, x_dirichlet;
    };

    __device__ __forceinline__ void
    role_setup(const DevGrid &g, const int I0, const int J0, PlaneRole &m)
    {
      m.row_exists = m.row_owned = m.row_dirichlet = 0u;
      m.J0       = J0;
      m.ny       = g.nc[1] * g.p;
      m.wrap_y   = g.wrap[1];
      m.lo_y     = g.ext_lo[1];
      m.stride_y = g.stride[1];
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int  Jr = m.wave + 4 * r, J = J0 + Jr;
          const bool ex = Jr <= 4 * TC && g.exists(1, J);
          m.row_exists |= (ex ? 1u : 0u) << r;
          m.row_owned |= ((ex && tile_owns(g, 1, Jr, J)) ? 1u : 0u) << r;
          m.row_dirichlet |= ((ex && g.dirichlet(1, J)) ? 1u : 0u) << r;
        }
      const int I     = I0 + m.lane;
      m.col_exists    = g.exists(0, I);
      m.col_off       = m.col_exists ? g.store_index(0, I) * g.stride[0] : 0;
      m.col_owned     = m.col_exists && tile_owns(g, 0, m.lane, I);
      m.col_dirichlet = m.col_exists && g.dirichlet(0, I);
      const int Jr = m.wave + 4 * m.lane, J = J0 + Jr, I64 = I0 + 4 * TC;
      m.x_exists    = m.lane <= TC && Jr <= 4 * TC && g.exists(0, I64) && g.exists(1, J);
      m.x_off       = m.x_exists ? g.store_index(0, I64) * g.stride[0] + g.store_index(1, J) * g.stride[1] : 0;
      m.x_owned     = m.x_exists && tile_owns(g, 0, 4 * TC, I64) && tile_owns(g, 1, Jr, J);
      m.x_dirichlet = m.x_exists && (g.dirichlet(0, I64) || g.dirichlet(1, J));
    }

    __device__ __forceinline__ long long
    row_offset(const PlaneRole &m, const int r)
    {
      int J = m.J0 + m.wave + 4 * r;
      if (m.wrap_y)
        J += (J < 0) ? m.ny : ((J >= m.ny) ? -m.ny : 0);
      else
        J -= m.lo_y;
      return J * m.stride_y;
    }

    // load plane K of src into registers (zero where nothing exists; MASK additionally
    // reads constrained DoFs as zero, as read_dof_values does for the operator)
    template <bool MASK>
    __device__ __forceinline__ void
    plane_load(const DevGrid &g, const PlaneRole &m, const double *__restrict__ src, const int K, const bool k_ok,
               double (&regs)[NR], double &extra)
    {
      const bool      kz   = k_ok && g.exists(2, K) && !(MASK && g.dirichlet(2, K));
      const double *  base = src + (kz ? g.store_index(2, K) * g.stride[2] : 0);
      const bool      cok  = kz && m.col_exists && !(MASK && m.col_dirichlet);
      const unsigned  rows = cok ? (MASK ? (m.row_exists & ~m.row_dirichlet) : m.row_exists) : 0u;
      const double *  col  = base + m.col_off;
#pragma unroll
      for (int r = 0; r < NR; ++r)
        regs[r] = ((rows >> r) & 1u) ? col[row_offset(m, r)] : 0.;
      extra = (kz && m.x_exists && !(MASK && m.x_dirichlet)) ? base[m.x_off] : 0.;
    }

    __device__ __forceinline__ void
    plane_to_lds(const PlaneRole &m, double *__restrict__ S, const double (&regs)[NR], const double extra)
    {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (m.wave + 4 * r <= 4 * TC)
          S[(m.wave + 4 * r) * PITCH + m.lane] = regs[r];
      if (m.lane <= TC && m.wave + 4 * m.lane <= 4 * TC)
        S[(m.wave + 4 * m.lane) * PITCH + 4 * TC] = extra;
    }

    // sum of the cells' contributions to one DoF, fixed order
    __device__ __forceinline__ double
    stage_out_sum(const double *__restrict__ S, const int Ir, const int Jr)
    {
      const int  cxa = Ir >> 2, i = Ir & 3, cya = Jr >> 2, j = Jr & 3;
      const bool xa = Ir < 4 * TC, xb = (i == 0) && (Ir > 0);
      const bool ya = Jr < 4 * TC, yb = (j == 0) && (Jr > 0);
      double     s  = 0.;
      if (xb && yb)
        s += S[(4 * 5 + 4) * SP + (cya - 1) * TC + cxa - 1];
      if (xa && yb)
        s += S[(4 * 5 + i) * SP + (cya - 1) * TC + cxa];
      if (xb && ya)
        s += S[(j * 5 + 4) * SP + cya * TC + cxa - 1];
      if (xa && ya)
        s += S[(j * 5 + i) * SP + cya * TC + cxa];
      return s;
    }

    // epilogue for one owned DoF: s = (P^-1 src)(dof) or (A src)(dof)
    template <int MODE>
    __device__ __forceinline__ void
    epilogue_store(const TiledArgs &a, const long long off, const double s, const bool constrained)
    {
      double t = s;
      if (MODE == OP_PLAIN || MODE == PC_PLAIN)
        a.dst[off] = constrained ? 0. : t;
      else if (MODE == OP_RESIDUAL)
        // (A x)_c = x_c on constrained DoFs (matrix_free_internal.h:247-255)
        a.dst[off] = a.b[off] - (constrained ? a.src[off] : t);
      else
        {
          if (constrained)
            t = a.src[off];
          if (MODE == PC_FIRST)
            a.dst[off] = a.f2 * t;
          else if (MODE == PC_STEP1)
            a.dst[off] = a.f1 * a.x[off] + a.f2 * t;
          else
            {
              const double xv = a.x[off];
              a.dst[off]      = xv + a.f1 * (xv - a.x_old[off]) + a.f2 * t;
            }
        }
    }

    template <int MODE>
    __device__ __forceinline__ void
    plane_assemble(const DevGrid &g, const PlaneRole &m, const TiledArgs &a, const double *__restrict__ S, const int K)
    {
      const long long offk = g.store_index(2, K) * g.stride[2];
      const bool      dk   = g.dirichlet(2, K);
      if (m.col_owned)
        {
          const long long offx = offk + m.col_off;
          const bool      dx   = dk || m.col_dirichlet;
          // rolled on purpose: the unrolled form keeps ~50 loads per thread in flight next to
          // the 250 registers of the cell tile and spills
#pragma unroll 4
          for (int r = 0; r < NR; ++r)
            if ((m.row_owned >> r) & 1u)
              epilogue_store<MODE>(a, offx + row_offset(m, r), stage_out_sum(S, m.lane, m.wave + 4 * r),
                                   dx || ((m.row_dirichlet >> r) & 1u));
        }
      if (m.x_owned)
        epilogue_store<MODE>(a, offk + m.x_off, stage_out_sum(S, 4 * TC, m.wave + 4 * m.lane), dk || m.x_dirichlet);
    }

    // -------------------------------------------------------------------------
    // FDM sweep
    // -------------------------------------------------------------------------
    constexpr int LDS_IN    = (4 * TC + 1) * PITCH; // raw input plane
    constexpr int LDS_OUT   = NE * SP;              // cell-major output plane
    constexpr int LDS_CARRY = NE * NT;              // per-cell top plane, private to its lane
    constexpr int LDS_EIG   = 16 * EIG;             // eigen tables (tensor-product instances)

    template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
    __global__ void __launch_bounds__(NT, 1)
    fdm_tiled_kernel(const DevGrid g, const DevFdm f, const TiledArgs a, const Tiling tl)
    {
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_in    = lds;
      double *S_out   = S_in + LDS_IN;
      double *S_carry = S_out + LDS_OUT;
      double *S_eig   = S_carry + LDS_CARRY;

      const int tid = threadIdx.x;
      const int b   = blockIdx.x;
      const int tx = b % tl.ntx, ty = (b / tl.ntx) % tl.nty, zc = b / (tl.ntx * tl.nty);
      const int X0 = tx * TO - 1, Y0 = ty * TO - 1; // tile cell (0,0)

      PlaneRole m;
      m.lane = tid & 63;
      m.wave = tid >> 6;
      role_setup(g, P * X0, P * Y0, m);

      // eigen tables of tensor-product meshes live in LDS
      const bool eig_in_lds = TENSOR && f.n_inst <= 16;
      if (eig_in_lds)
        for (int e = tid; e < f.n_inst * EIG; e += NT)
          S_eig[e] = f.eig[e];

      // compute role: this lane's cell
      const int  cxl = tid % TC, cyl = tid / TC;
      const int  cx = X0 + cxl, cy = Y0 + cyl;
      const bool cell_xy = cell_computable(g, 0, cx) && cell_computable(g, 1, cy);
      const int  Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int  z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;
      const double *in_cell = S_in + (4 * cyl) * PITCH + 4 * cxl;

      double t[N1 * NE]; // the cell tile, index c*25 + b*5 + a
#pragma unroll
      for (int e = 0; e < NE; ++e)
        S_carry[e * NT + tid] = 0.;

      double pre[NR], pre_x; // prefetched plane
      plane_load<false>(g, m, a.src, P * z_first, true, pre, pre_x);

      for (int cz = z_first; cz < Ze; ++cz)
        {
          const bool active = cell_xy;
          int ix = 0, iy = 0, iz = 0;
          if (active)
            {
              const long long ce = (cx + 1) + (long long)(g.nc[0] + 2) * ((cy + 1) + (long long)(g.nc[1] + 2) * (cz + 1));
              ix = TENSOR ? f.inst[0][cx + 1] : f.inst[0][ce];
              iy = TENSOR ? f.inst[1][cy + 1] : f.inst[1][ce];
              iz = TENSOR ? f.inst[2][cz + 1] : f.inst[2][ce];
            }
          if (TENSOR)
            iz = f.inst[2][cz + 1]; // wave-uniform: the z matrices become scalar operands
          const double *etab = eig_in_lds ? S_eig : f.eig;
          const double *ex = etab + ix * EIG, *ey = etab + iy * EIG;
          const double *ez = (TENSOR ? f.eig : etab) + iz * EIG;
          double wpx[N1], wpy[N1];
          if (HAS_PRE)
            {
#pragma unroll
              for (int i = 0; i < N1; ++i)
                {
                  wpx[i] = f.w_pre[0][(cx + 1) * N1 + i];
                  wpy[i] = f.w_pre[1][(cy + 1) * N1 + i];
                }
            }

          // ---- forward: stream the five src planes of this cell layer ------------
#pragma unroll 1
          for (int k = 0; k < N1; ++k)
            {
              __syncthreads(); // S_in free
              plane_to_lds(m, S_in, pre, pre_x);
              __syncthreads();
              // prefetch the next plane: k+1 of this layer, or plane 0 of the next layer
              {
                const int  Kn   = (k < N1 - 1) ? P * cz + k + 1 : P * (cz + 1);
                const bool more = (k < N1 - 1) || (cz + 1 < Ze);
                plane_load<false>(g, m, a.src, Kn, more, pre, pre_x);
              }
              if (active)
                {
                  double v[NE];
#pragma unroll
                  for (int j = 0; j < N1; ++j)
#pragma unroll
                    for (int i = 0; i < N1; ++i)
                      v[j * 5 + i] = in_cell[j * PITCH + i];
                  if (HAS_PRE)
                    {
                      const double wz = f.w_pre[2][(cz + 1) * N1 + k];
#pragma unroll
                      for (int j = 0; j < N1; ++j)
#pragma unroll
                        for (int i = 0; i < N1; ++i)
                          v[j * 5 + i] *= wpx[i] * (wpy[j] * wz);
                    }
                  double q[NE];
                  load_q(q, ex);
                  plane_x<true>(v, q);
                  load_q(q, ey);
                  plane_y<true>(v, q);
                  // park the x-y transformed plane; the z sweeps run after the fifth plane
#pragma unroll
                  for (int kk = 0; kk < N1; ++kk)
                    if (kk == k)
                      {
#pragma unroll
                        for (int e = 0; e < NE; ++e)
                          t[kk * NE + e] = v[e];
                      }
                }
            }

          // ---- z: forward sweep, scale by 1/(lx + ly + lz), backward sweep, line by line ----
          if (active)
            {
              double qz[NE];
              load_q(qz, ez);
              const double *inv = f.inv_table != nullptr ?
                                    f.inv_table + (long long)((iz * f.n_inst + iy) * f.n_inst + ix) * (N1 * NE) :
                                    nullptr;
              double lx[N1], ly[N1], lz[N1];
              if (inv == nullptr)
                {
#pragma unroll
                  for (int i = 0; i < N1; ++i)
                    {
                      lx[i] = ex[NE + i];
                      ly[i] = ey[NE + i];
                      lz[i] = ez[NE + i];
                    }
                }
#pragma unroll
              for (int e = 0; e < NE; ++e)
                {
                  line5<true>(t[0 * NE + e], t[1 * NE + e], t[2 * NE + e], t[3 * NE + e], t[4 * NE + e], qz);
                  if (inv != nullptr)
                    {
#pragma unroll
                      for (int c = 0; c < N1; ++c)
                        t[c * NE + e] *= inv[c * NE + e];
                    }
                  else
                    {
#pragma unroll
                      for (int c = 0; c < N1; ++c)
                        {
                          // reciprocal with two Newton steps (full double precision)
                          const double sm = lx[e % 5] + (ly[e / 5] + lz[c]);
                          double       r  = __builtin_amdgcn_rcp(sm);
                          r               = fma(fma(-sm, r, 1.0), r, r);
                          r               = fma(fma(-sm, r, 1.0), r, r);
                          t[c * NE + e] *= r;
                        }
                    }
                  line5<false>(t[0 * NE + e], t[1 * NE + e], t[2 * NE + e], t[3 * NE + e], t[4 * NE + e], qz);
                }
            }

          // ---- backward in y, x per output plane; exchange; owners store ----------
          const bool halo_layer = cz < Zb; // only its top plane (k = 4) is needed
          double     wqx[N1], wqy[N1];
          if (HAS_POST)
            {
#pragma unroll
              for (int i = 0; i < N1; ++i)
                {
                  wqx[i] = f.w_post[0][(cx + 1) * N1 + i];
                  wqy[i] = f.w_post[1][(cy + 1) * N1 + i];
                }
            }
#pragma unroll
          for (int k = 0; k < N1; ++k)
            {
              if (halo_layer && k < N1 - 1)
                continue;
              double o[NE];
              if (active)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    o[e] = t[k * NE + e];
                  double q[NE];
                  load_q(q, ey);
                  plane_y<false>(o, q);
                  load_q(q, ex);
                  plane_x<false>(o, q);
                  if (HAS_POST)
                    {
                      const double wz = f.w_post[2][(cz + 1) * N1 + k];
#pragma unroll
                      for (int j = 0; j < N1; ++j)
#pragma unroll
                        for (int i = 0; i < N1; ++i)
                          o[j * 5 + i] *= wqx[i] * (wqy[j] * wz);
                    }
                }
              else
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    o[e] = 0.;
                }
              if (k == N1 - 1)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    S_carry[e * NT + tid] = o[e];
                  break;
                }
              if (k == 0)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    o[e] += S_carry[e * NT + tid];
                }
              __syncthreads(); // S_out free
#pragma unroll
              for (int e = 0; e < NE; ++e)
                S_out[e * SP + tid] = o[e];
              __syncthreads();
              plane_assemble<MODE>(g, m, a, S_out, P * cz + k);
            }
        }
      // top plane of the block (physical boundary): only the last layer contributes
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        {
          __syncthreads();
#pragma unroll
          for (int e = 0; e < NE; ++e)
            S_out[e * SP + tid] = S_carry[e * NT + tid];
          __syncthreads();
          plane_assemble<MODE>(g, m, a, S_out, P * g.nc[2]);
        }
    }

    Tiling
    make_tiling(const DevGrid &g)
    {
      Tiling tl;
      tl.ntx = (g.nc[0] + TO - 1) / TO;
      tl.nty = (g.nc[1] + TO - 1) / TO;
      // the last owned plane nc*p of a non-wrapped direction is produced by the tile
      // that holds cell nc-1, no extra tile needed
      const int tiles = tl.ntx * tl.nty;
      // one workgroup per CU (104 KB of LDS each): aim at ~256 workgroups, chunks of >= 2 layers
      int best = std::max(1, std::min(256 / tiles, g.nc[2] / 2));
      best     = std::min(best, g.nc[2]);
      tl.zlen = (g.nc[2] + best - 1) / best;
      tl.nzc  = (g.nc[2] + tl.zlen - 1) / tl.zlen;
      return tl;
    }
  } // namespace

  bool
  laplace_tiled(const DevGrid &, const DevOp &, const TiledArgs &, hipStream_t)
  {
    return false;
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST, bool TENSOR>
  static void
  launch_fdm_tiled_t(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    const size_t lds  = (LDS_IN + LDS_OUT + LDS_CARRY + LDS_EIG) * sizeof(double);
    static bool  done = false;
    if (!done)
      {
        (void)hipFuncSetAttribute((const void *)fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
      }
    hipLaunchKernelGGL((fdm_tiled_kernel<MODE, HAS_PRE, HAS_POST, TENSOR>), dim3(tl.ntx * tl.nty * tl.nzc), dim3(NT), lds, st, g, f, a,
                       tl);
  }

  template <int MODE, bool HAS_PRE, bool HAS_POST>
  static void
  launch_fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, const Tiling &tl, hipStream_t st)
  {
    if (f.tensor_inst)
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
