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
    // column Ir = lane and rows Jr = wave + 4 r; lanes <= 16 of every wave also
    // handle column Ir = 64, rows Jr = wave + 4 lane.
    // -------------------------------------------------------------------------
    struct PlaneMap
    {
      int I0, J0;   // logical coordinates of tile-relative DoF (0,0)
      int lane, wave;
    };

    // scatter one DoF value into the staging slots of the cells that contain it
    __device__ __forceinline__ void
    stage_in_write(double *__restrict__ S, const int Ir, const int Jr, const double val)
    {
      const int  cxa = Ir >> 2, i = Ir & 3, cya = Jr >> 2, j = Jr & 3;
      const bool xa = Ir < 4 * TC, xb = (i == 0) && (Ir > 0);
      const bool ya = Jr < 4 * TC, yb = (j == 0) && (Jr > 0);
      if (xa && ya)
        S[(j * 5 + i) * SP + cya * TC + cxa] = val;
      if (xb && ya)
        S[(j * 5 + 4) * SP + cya * TC + cxa - 1] = val;
      if (xa && yb)
        S[(4 * 5 + i) * SP + (cya - 1) * TC + cxa] = val;
      if (xb && yb)
        S[(4 * 5 + 4) * SP + (cya - 1) * TC + cxa - 1] = val;
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

    // is tile-relative DoF coordinate Ir (logical I) written by this tile?
    __device__ __forceinline__ bool
    tile_owns(const DevGrid &g, const int d, const int Ir, const int I)
    {
      if (I < 0 || I > g.own_hi[d] || Ir < P)
        return false;
      return Ir < 4 * TC || I == g.nc[d] * P;
    }

    // load plane K of src for the whole tile into registers (zero where nothing exists;
    // MASK_DIRICHLET additionally reads constrained DoFs as zero)
    template <bool MASK_DIRICHLET>
    __device__ __forceinline__ void
    plane_load(const DevGrid &g, const PlaneMap &m, const double *__restrict__ src, const int K, const bool k_ok,
               double (&regs)[NR], double &extra)
    {
      const bool kz   = k_ok && g.exists(2, K) && !(MASK_DIRICHLET && g.dirichlet(2, K));
      const long long offk = kz ? g.store_index(2, K) * g.stride[2] : 0;
      const int  I    = m.I0 + m.lane;
      const bool xok  = kz && g.exists(0, I) && !(MASK_DIRICHLET && g.dirichlet(0, I));
      const long long offx = xok ? offk + g.store_index(0, I) * g.stride[0] : 0;
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int  Jr = m.wave + 4 * r;
          const int  J  = m.J0 + Jr;
          const bool ok = xok && Jr <= 4 * TC && g.exists(1, J) && !(MASK_DIRICHLET && g.dirichlet(1, J));
          regs[r]       = ok ? src[offx + g.store_index(1, J) * g.stride[1]] : 0.;
        }
      {
        // column Ir = 64
        const int  Jr = m.wave + 4 * m.lane;
        const int  J = m.J0 + Jr, I64 = m.I0 + 4 * TC;
        const bool ok = kz && m.lane <= TC && Jr <= 4 * TC && g.exists(0, I64) && g.exists(1, J) &&
                        !(MASK_DIRICHLET && (g.dirichlet(0, I64) || g.dirichlet(1, J)));
        extra = ok ? src[offk + g.store_index(0, I64) * g.stride[0] + g.store_index(1, J) * g.stride[1]] : 0.;
      }
    }

    __device__ __forceinline__ void
    plane_stage(const PlaneMap &m, double *__restrict__ S, const double (&regs)[NR], const double extra)
    {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int Jr = m.wave + 4 * r;
          if (Jr <= 4 * TC)
            stage_in_write(S, m.lane, Jr, regs[r]);
        }
      const int Jr = m.wave + 4 * m.lane;
      if (m.lane <= TC && Jr <= 4 * TC)
        stage_in_write(S, 4 * TC, Jr, extra);
    }

    // epilogue for one owned DoF: s = (P^-1 src)(dof) or (A src)(dof)
    __device__ __forceinline__ void
    epilogue_store(const TiledArgs &a, const long long off, const double s, const bool constrained)
    {
      double t = s;
      switch (a.mode)
        {
          case OP_PLAIN:
          case PC_PLAIN:
            a.dst[off] = constrained ? 0. : t;
            break;
          case OP_RESIDUAL:
            // (A x)_c = x_c on constrained DoFs (matrix_free_internal.h:247-255)
            a.dst[off] = a.b[off] - (constrained ? a.src[off] : t);
            break;
          case PC_FIRST:
            if (constrained)
              t = a.src[off];
            a.dst[off] = a.f2 * t;
            break;
          case PC_STEP1:
            if (constrained)
              t = a.src[off];
            a.dst[off] = a.f1 * a.x[off] + a.f2 * t;
            break;
          default: // PC_UPDATE
            {
              if (constrained)
                t = a.src[off];
              const double xv = a.x[off];
              a.dst[off]      = xv + a.f1 * (xv - a.x_old[off]) + a.f2 * t;
            }
        }
    }

    __device__ __forceinline__ void
    plane_assemble(const DevGrid &g, const PlaneMap &m, const TiledArgs &a, const double *__restrict__ S, const int K)
    {
      const long long offk = g.store_index(2, K) * g.stride[2];
      const bool      dk   = g.dirichlet(2, K);
      {
        const int I = m.I0 + m.lane;
        if (tile_owns(g, 0, m.lane, I))
          {
            const long long offx = offk + g.store_index(0, I) * g.stride[0];
            const bool      dx   = dk || g.dirichlet(0, I);
#pragma unroll
            for (int r = 0; r < NR; ++r)
              {
                const int Jr = m.wave + 4 * r;
                const int J  = m.J0 + Jr;
                if (Jr <= 4 * TC && tile_owns(g, 1, Jr, J))
                  epilogue_store(a, offx + g.store_index(1, J) * g.stride[1], stage_out_sum(S, m.lane, Jr),
                                 dx || g.dirichlet(1, J));
              }
          }
      }
      {
        const int Jr = m.wave + 4 * m.lane;
        const int J = m.J0 + Jr, I64 = m.I0 + 4 * TC;
        if (m.lane <= TC && Jr <= 4 * TC && tile_owns(g, 0, 4 * TC, I64) && tile_owns(g, 1, Jr, J))
          epilogue_store(a, offk + g.store_index(0, I64) * g.stride[0] + g.store_index(1, J) * g.stride[1],
                         stage_out_sum(S, 4 * TC, Jr), dk || g.dirichlet(0, I64) || g.dirichlet(1, J));
      }
    }

    // -------------------------------------------------------------------------
    // FDM sweep
    // -------------------------------------------------------------------------
    __global__ void __launch_bounds__(NT, 1)
    fdm_tiled_kernel(const DevGrid g, const DevFdm f, const TiledArgs a, const Tiling tl)
    {
      extern __shared__ __attribute__((aligned(16))) double lds[];
      double *S_in  = lds;
      double *S_out = lds + NE * SP;

      const int tid = threadIdx.x;
      int       b   = blockIdx.x;
      const int tx = b % tl.ntx, ty = (b / tl.ntx) % tl.nty, zc = b / (tl.ntx * tl.nty);

      PlaneMap m;
      m.lane = tid & 63;
      m.wave = tid >> 6;
      const int X0 = tx * TO - 1, Y0 = ty * TO - 1; // tile cell (0,0)
      m.I0 = P * X0;
      m.J0 = P * Y0;

      // compute role: this lane's cell
      const int  cx = X0 + (tid % TC), cy = Y0 + (tid / TC);
      const bool cell_xy = cell_computable(g, 0, cx) && cell_computable(g, 1, cy);
      const int Zb = zc * tl.zlen, Ze = min(Zb + tl.zlen, g.nc[2]);
      const int z_first = cell_computable(g, 2, Zb - 1) ? Zb - 1 : Zb;

      // per-lane tables
      const double *wpx = f.w_pre[0] + (cx + 1) * N1, *wpy = f.w_pre[1] + (cy + 1) * N1;
      const double *wqx = f.w_post[0] + (cx + 1) * N1, *wqy = f.w_post[1] + (cy + 1) * N1;

      double t[N1 * NE]; // the cell tile, index c*25 + b*5 + a
      double carry[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e)
        carry[e] = 0.;

      double pre[NR], pre_x; // prefetched plane
      plane_load<false>(g, m, a.src, P * z_first, true, pre, pre_x);

      for (int cz = z_first; cz < Ze; ++cz)
        {
          const bool active = cell_xy; // layer cz is computable by construction
          const long long ce =
            active ? (cx + 1) + (long long)(g.nc[0] + 2) * ((cy + 1) + (long long)(g.nc[1] + 2) * (cz + 1)) : 0;
          const double *ex = f.eig, *ey = f.eig, *ez = f.eig;
          if (active)
            {
              ex += (long long)(f.tensor_inst ? f.inst[0][cx + 1] : f.inst[0][ce]) * EIG;
              ey += (long long)(f.tensor_inst ? f.inst[1][cy + 1] : f.inst[1][ce]) * EIG;
              ez += (long long)(f.tensor_inst ? f.inst[2][cz + 1] : f.inst[2][ce]) * EIG;
            }
          const double *wpz = f.w_pre[2] + (cz + 1) * N1, *wqz = f.w_post[2] + (cz + 1) * N1;

#pragma unroll
          for (int e = 0; e < N1 * NE; ++e)
            t[e] = 0.;

          // ---- forward: stream the five src planes of this cell layer ------------
#pragma unroll 1
          for (int k = 0; k < N1; ++k)
            {
              __syncthreads(); // S_in free
              plane_stage(m, S_in, pre, pre_x);
              __syncthreads();
              // prefetch the next plane: k+1 of this layer, or plane 0 of the next layer
              {
                const int  Kn = (k < N1 - 1) ? P * cz + k + 1 : P * (cz + 1);
                const bool more = (k < N1 - 1) || (cz + 1 < Ze);
                plane_load<false>(g, m, a.src, Kn, more, pre, pre_x);
              }
              if (active)
                {
                  double v[NE];
                  const double wz = wpz[k];
#pragma unroll
                  for (int j = 0; j < N1; ++j)
#pragma unroll
                    for (int i = 0; i < N1; ++i)
                      v[j * 5 + i] = S_in[(j * 5 + i) * SP + tid] * (wpx[i] * wpy[j] * wz);
                  double q[NE];
                  load_q(q, ex);
                  plane_x<true>(v, q);
                  load_q(q, ey);
                  plane_y<true>(v, q);
                  // z forward: t[c] += Qz[k][c] v
                  const double q0 = ez[k * N1 + 0], q1 = ez[k * N1 + 1], q2 = ez[k * N1 + 2], q3 = ez[k * N1 + 3],
                               q4 = ez[k * N1 + 4];
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    {
                      t[0 * NE + e] = fma(q0, v[e], t[0 * NE + e]);
                      t[1 * NE + e] = fma(q1, v[e], t[1 * NE + e]);
                      t[2 * NE + e] = fma(q2, v[e], t[2 * NE + e]);
                      t[3 * NE + e] = fma(q3, v[e], t[3 * NE + e]);
                      t[4 * NE + e] = fma(q4, v[e], t[4 * NE + e]);
                    }
                }
            }

          // ---- scale by 1/(lx + ly + lz), backward in z ---------------------------
          if (active)
            {
              double lx[N1], ly[N1], lz[N1];
#pragma unroll
              for (int i = 0; i < N1; ++i)
                {
                  lx[i] = ex[NE + i];
                  ly[i] = ey[NE + i];
                  lz[i] = ez[NE + i];
                }
#pragma unroll
              for (int c = 0; c < N1; ++c)
#pragma unroll
                for (int bb = 0; bb < N1; ++bb)
#pragma unroll
                  for (int aa = 0; aa < N1; ++aa)
                    t[c * NE + bb * 5 + aa] = t[c * NE + bb * 5 + aa] / (lx[aa] + (ly[bb] + lz[c]));
              double q[NE];
              load_q(q, ez);
#pragma unroll
              for (int e = 0; e < NE; ++e)
                line5<false>(t[0 * NE + e], t[1 * NE + e], t[2 * NE + e], t[3 * NE + e], t[4 * NE + e], q);
            }

          // ---- backward in y, x per output plane; exchange; owners store ----------
          const bool halo_layer = cz < Zb; // only its top plane (k = 4) is needed
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
                  const double wz = wqz[k];
#pragma unroll
                  for (int j = 0; j < N1; ++j)
#pragma unroll
                    for (int i = 0; i < N1; ++i)
                      o[j * 5 + i] *= wqx[i] * wqy[j] * wz;
                }
              else
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    o[e] = 0.;
                }
              if (k == 0)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    o[e] += carry[e];
                }
              if (k == N1 - 1)
                {
#pragma unroll
                  for (int e = 0; e < NE; ++e)
                    carry[e] = o[e];
                  break;
                }
              __syncthreads(); // S_out free
#pragma unroll
              for (int e = 0; e < NE; ++e)
                S_out[e * SP + tid] = o[e];
              __syncthreads();
              plane_assemble(g, m, a, S_out, P * cz + k);
            }
        }
      // top plane of the block (physical boundary): only the last layer contributes
      if (Ze == g.nc[2] && g.own_hi[2] == g.nc[2] * P)
        {
          __syncthreads();
#pragma unroll
          for (int e = 0; e < NE; ++e)
            S_out[e * SP + tid] = carry[e];
          __syncthreads();
          plane_assemble(g, m, a, S_out, P * g.nc[2]);
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

  bool
  fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, hipStream_t st)
  {
    if (g.p != P || f.m != N1 || f.vertex)
      return false;
    const Tiling tl     = make_tiling(g);
    const size_t lds    = 2 * NE * SP * sizeof(double);
    static bool  attr_set = false;
    if (!attr_set)
      {
        (void)hipFuncSetAttribute((const void *)fdm_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
      }
    hipLaunchKernelGGL(fdm_tiled_kernel, dim3(tl.ntx * tl.nty * tl.nzc), dim3(NT), lds, st, g, f, a, tl);
    return true;
  }
} // namespace asmamd
