// Prototype of the "one lane per cell" sweep (round 4): a wave owns a TX x TY tile of cells (64 lanes) and marches along
// z; a lane keeps the whole 5x5x5 tile of its cell in registers, so all three sweep directions are register FMA chains:
// no LDS transposes, no workgroup barrier, waves are independent.  The DoF plane two consecutive cells of a column share
// is transformed once in x-y (forward) and once back (the two cells' contributions are added in the transformed space).
// Uniform interior instance everywhere, even-odd sweeps with scalar operands; dst = x + f2 * (sum of cell results).
// Checks itself against a naive one-thread-per-cell kernel with atomics, then times the sweep.
//   hipcc -O3 --offload-arch=gfx950 -o proto_lane proto_lane.hip ; ./proto_lane [nx ny nz] [zchunks]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

constexpr int P = 4, N1 = 5, NE = 25;
#ifndef TXV
#  define TXV 8
#endif
constexpr int TX = TXV, TY = 64 / TX, OX = TX - 1, OY = TY - 1;

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef dbl2   dbl2_u __attribute__((aligned(8)));

struct Args
{
  const double *src, *x;
  double *      dst;
  double        f2;
  int           nc[3];
  int           sy;      // row pitch (doubles)
  long long     sz;      // plane pitch
  const double *QF, *QB; // [i][a] forward (with pre weights) / backward (with post weights) matrices, columns 0..2 even, 3..4 odd
  const double *QFc, *QBc; // the same in compact even-odd form (EoMatrix)
  const double *D;       // [(b*5+a)*5 + c]
  int           ntx, nty, nzc, zlen;
};

// a matrix with even columns 0..2 and odd columns 3, 4 needs 13 numbers: E[a][i] = q[i][a] (a, i < 3), O[a][i] = q[i][3 + a]
// (a, i < 2); stored as 16 doubles so that two scalar loads fetch it.  Wave-uniform: scalar operands of the FMAs.
struct EoMatrix
{
  const __attribute__((address_space(4))) double *p;
  __device__ __forceinline__ explicit EoMatrix(const double *q)
    : p((const __attribute__((address_space(4))) double *)q)
  {}
  __device__ __forceinline__ double E(const int a, const int i) const { return p[a * 3 + i]; }
  __device__ __forceinline__ double O(const int a, const int i) const { return p[9 + a * 2 + i]; }
  // the same table behind a pointer the compiler cannot trace: loaded where the copy is made
  __device__ __forceinline__ EoMatrix fresh() const
  {
    EoMatrix r(*this);
    asm volatile("" : "+s"(r.p));
    return r;
  }
};
// the 13 numbers in scalar registers
struct EoRegs
{
  double e[9], o[4];
  __device__ __forceinline__ explicit EoRegs(const EoMatrix &m)
  {
#pragma unroll
    for (int k = 0; k < 9; ++k)
      e[k] = m.p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = m.p[9 + k];
  }
  __device__ __forceinline__ double E(const int a, const int i) const { return e[a * 3 + i]; }
  __device__ __forceinline__ double O(const int a, const int i) const { return o[a * 2 + i]; }
};

// the 13 numbers in vector registers, read from an LDS copy of the table (uniform address: broadcast)
struct EoVregs
{
  double e[9], o[4];
  __device__ __forceinline__ explicit EoVregs(const double *m)
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

template <bool FWD, typename Q>
__device__ __forceinline__ void
line5_eo(double &u0, double &u1, double &u2, double &u3, double &u4, const Q &q)
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

// x and y sweeps of one plane v[j*5 + i]
template <bool FWD, typename Q>
__device__ __forceinline__ void
plane2d(double (&v)[NE], const Q &qx, const Q &qy)
{
  if (FWD)
    {
#pragma unroll
      for (int j = 0; j < N1; ++j)
        line5_eo<true>(v[j * 5 + 0], v[j * 5 + 1], v[j * 5 + 2], v[j * 5 + 3], v[j * 5 + 4], qx);
#pragma unroll
      for (int i = 0; i < N1; ++i)
        line5_eo<true>(v[0 * 5 + i], v[1 * 5 + i], v[2 * 5 + i], v[3 * 5 + i], v[4 * 5 + i], qy);
    }
  else
    {
#pragma unroll
      for (int i = 0; i < N1; ++i)
        line5_eo<false>(v[0 * 5 + i], v[1 * 5 + i], v[2 * 5 + i], v[3 * 5 + i], v[4 * 5 + i], qy);
#pragma unroll
      for (int j = 0; j < N1; ++j)
        line5_eo<false>(v[j * 5 + 0], v[j * 5 + 1], v[j * 5 + 2], v[j * 5 + 3], v[j * 5 + 4], qx);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-pair form.  A workgroup = two waves that share one tile of TX x TY cells (lane l of both waves = cell l).  Per
// cell layer wave r transforms two of the four new DoF planes in x-y (planes 1 + 2r, 2 + 2r), both write their planes
// to an LDS scratchpad (lane-strided: conflict-free), each takes 13 of the 25 z columns through the z sweeps
// (in place in the scratchpad) and transforms two of the four output planes back (planes 2r, 2r + 1), merges them with
// the neighbouring cells and stores them with the fused update.  Nothing is parked in registers; the accumulation
// registers hold the next layer's source planes and update operands, requested a whole layer ahead.
// ---------------------------------------------------------------------------------------------------------------
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));

struct Lane
{
  int  cx, cy;
  // cell_xy: the cell is computed.  The cells cx = nc / cy = nc do not exist but own the block's last DoF column / row
  // (their column 0 / row 0): they run along with zero results (zero scaling table) and store what they receive.
  bool cell_xy, owned, left_ok, below_ok, ph_x, ph_y;
  int  off0;          // storage offset of the cell's DoF (0,0) inside a plane
  unsigned row8[N1];  // source loads: byte offset of the cell's DoF row j inside a plane (cells that are not computed: cell 0)
  unsigned xrow8[P];  // operand loads: the same for the DoFs the lane may own (rows that do not exist: row 0)
};

// The next layer's source planes and update operands wait in FIXED accumulation registers (the compiler never sees
// them: it must not use accumulation registers itself - audit: no v_accvgpr_* outside the asm statements, no spills):
//   a[42..91], a[92..141]    the two source planes of this wave: row j at +10 j: x 0..1, x 2..3, x 4
//   a[142..173], a[174..205] the update operands of its two output planes: row j at +8 j
//   a[206..255]              prologue: plane 0 of the first layer
// (the registers below a42 are left to the compiler, which takes them from a0 upwards if it has to spill)
constexpr int A_S0 = 42, A_S1 = 92, A_X0 = 142, A_X1 = 174, A_P0 = 206, A_END = 256;
#define A_CLOBBER_16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void
areg_reserve()
{
  asm volatile("" ::: "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", A_CLOBBER_16(5), A_CLOBBER_16(6), A_CLOBBER_16(7), A_CLOBBER_16(8), A_CLOBBER_16(9),
               A_CLOBBER_16(10), A_CLOBBER_16(11), A_CLOBBER_16(12), A_CLOBBER_16(13), A_CLOBBER_16(14), A_CLOBBER_16(15), A_CLOBBER_16(16), A_CLOBBER_16(17),
               A_CLOBBER_16(18), A_CLOBBER_16(19), A_CLOBBER_16(20), A_CLOBBER_16(21), A_CLOBBER_16(22), A_CLOBBER_16(23), A_CLOBBER_16(24), "a250", "a251",
               "a252", "a253", "a254", "a255");
}

template <int A0, int J>
__device__ __forceinline__ void
plane_request_row(const double *plane, const Lane &c)
{
  const unsigned o = c.row8[J];
  asm volatile("global_load_dwordx4 a[%2:%3], %0, %1\n\t"
               "global_load_dwordx4 a[%4:%5], %0, %1 offset:16\n\t"
               "global_load_dwordx2 a[%6:%7], %0, %1 offset:32" ::"v"(o),
               "s"(plane), "n"(A0 + 10 * J), "n"(A0 + 10 * J + 3), "n"(A0 + 10 * J + 4), "n"(A0 + 10 * J + 7), "n"(A0 + 10 * J + 8), "n"(A0 + 10 * J + 9)
               : "memory");
}
template <int A0>
__device__ __forceinline__ void
plane_request(const double *plane, const Lane &c)
{
  plane_request_row<A0, 0>(plane, c);
  plane_request_row<A0, 1>(plane, c);
  plane_request_row<A0, 2>(plane, c);
  plane_request_row<A0, 3>(plane, c);
  plane_request_row<A0, 4>(plane, c);
}
template <int A0, int J>
__device__ __forceinline__ void
x_request_row(const double *plane, const Lane &c)
{
  const unsigned o = c.xrow8[J];
  asm volatile("global_load_dwordx4 a[%2:%3], %0, %1\n\t"
               "global_load_dwordx4 a[%4:%5], %0, %1 offset:16" ::"v"(o),
               "s"(plane), "n"(A0 + 8 * J), "n"(A0 + 8 * J + 3), "n"(A0 + 8 * J + 4), "n"(A0 + 8 * J + 7)
               : "memory");
}
template <int A0>
__device__ __forceinline__ void
x_request(const double *plane, const Lane &c)
{
  x_request_row<A0, 0>(plane, c);
  x_request_row<A0, 1>(plane, c);
  x_request_row<A0, 2>(plane, c);
  x_request_row<A0, 3>(plane, c);
}
// everything but the N youngest vector memory operations has completed
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
template <int A0, int... E>
__device__ __forceinline__ void
aregs_to(double *v, std::integer_sequence<int, E...>)
{
  // element E of a block of doubles stored at a[A0 + 2 E]
  ((v[E] = areg_double<A0 + 2 * E>()), ...);
}
template <int A0>
__device__ __forceinline__ void
plane_unpack(double (&v)[NE])
{
  // row j at A0 + 10 j: five doubles
  aregs_to<A0 + 0>(&v[0], std::make_integer_sequence<int, 5>());
  aregs_to<A0 + 10>(&v[5], std::make_integer_sequence<int, 5>());
  aregs_to<A0 + 20>(&v[10], std::make_integer_sequence<int, 5>());
  aregs_to<A0 + 30>(&v[15], std::make_integer_sequence<int, 5>());
  aregs_to<A0 + 40>(&v[20], std::make_integer_sequence<int, 5>());
}
template <int A0>
__device__ __forceinline__ void
x_unpack(double (&x)[P * P])
{
  aregs_to<A0>(&x[0], std::make_integer_sequence<int, P * P>());
}

// value of lane - 1 inside a row of 16 lanes (DPP row_shr:1; the first lane of a row reads 0)
__device__ __forceinline__ double
lane_left(const double x)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// merges in x and y, fused update and store of the DoFs of plane K the lane owns; o[j*5+i] is consumed
__device__ __forceinline__ void
emit_plane(const Args &a, const Lane &c, const int K, double (&o)[NE], const double (&xa)[P * P])
{
  // x: column 4 of the left cell -> column 0 (cells that are not computed hold zeros)
#pragma unroll
  for (int j = 0; j < N1; ++j)
    {
      const double r = lane_left(o[j * 5 + 4]);
      if (c.left_ok)
        o[j * 5 + 0] += r;
    }
  // y: row 4 of the cell below -> row 0
  double rb[P];
#pragma unroll
  for (int i = 0; i < P; ++i)
    rb[i] = __shfl_up(o[4 * 5 + i], TX, 64);
  if (c.below_ok)
    {
#pragma unroll
      for (int i = 0; i < P; ++i)
        o[0 * 5 + i] += rb[i];
    }
  if (c.owned)
    {
      double *const d = a.dst + (long long)K * a.sz + c.off0;
#pragma unroll
      for (int j = 0; j < P; ++j)
        if (j == 0 || !c.ph_y)
          {
            dbl2 lo, hi;
            lo.x = fma(a.f2, o[j * 5 + 0], xa[j * P + 0]);
            lo.y = fma(a.f2, o[j * 5 + 1], xa[j * P + 1]);
            hi.x = fma(a.f2, o[j * 5 + 2], xa[j * P + 2]);
            hi.y = fma(a.f2, o[j * 5 + 3], xa[j * P + 3]);
#if ABL & 4
            if (lo.x != 1.2345e300)
              continue;
#endif
            if (c.ph_x)
              __builtin_nontemporal_store(lo.x, d + j * a.sy);
            else
              {
#if ABL & 32
                *(dbl2_u *)(d + j * a.sy)     = lo;
                *(dbl2_u *)(d + j * a.sy + 2) = hi;
#else
                __builtin_nontemporal_store(lo, (dbl2_u *)(d + j * a.sy));
                __builtin_nontemporal_store(hi, (dbl2_u *)(d + j * a.sy + 2));
#endif
              }
          }
    }
}

constexpr int NZC  = 13;                 // z columns per wave: 0..12 / 13..24 (the second wave's last line is idle)
constexpr int XLDS = P * NE * 64;        // doubles of the scratchpad: slot (q, e) of lane l at (q * 25 + e) * 64 + l

__device__ unsigned long long g_stamp[8];
__device__ unsigned long long g_life[4096][2];
__device__ unsigned           g_hw[4096][2];
#define STAMP(i)                                                     \
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
#ifndef ABL
#  define ABL 0
#endif

__global__ void __launch_bounds__(128) lane_kernel(const Args a)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const bool               stamp_on = blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x < 64;
  unsigned long long       stamp_t = __builtin_amdgcn_s_memtime(), stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long life_t0 = __builtin_amdgcn_s_memrealtime();
  // work index -> (tile x, tile y, z chunk); blocks b and b + 8 share an XCD: contiguous ranges per XCD
  const int n = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int wi = xcd * (n >> 3) + min(xcd, n & 7) + idx;
  const int tx = wi % a.ntx, ty = (wi / a.ntx) % a.nty, zc = wi / (a.ntx * a.nty);
  const int X0 = tx * OX, Y0 = ty * OY;

  const int lane = threadIdx.x & 63, cl = lane % TX, rw = lane / TX;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 / 1, wave-uniform
  Lane      c;
  c.cx       = X0 + cl;
  c.cy       = Y0 + rw;
  c.cell_xy         = c.cx < a.nc[0] && c.cy < a.nc[1];
  const bool exists = c.cx <= a.nc[0] && c.cy <= a.nc[1];
  c.ph_x            = c.cx == a.nc[0];
  c.ph_y            = c.cy == a.nc[1];
  c.owned           = exists && (cl >= 1 || tx == 0) && (rw >= 1 || ty == 0);
  c.left_ok         = exists && cl >= 1;
  c.below_ok        = exists && rw >= 1;
  c.off0            = exists ? (P * c.cy) * a.sy + P * c.cx : 0;
#pragma unroll
  for (int j = 0; j < N1; ++j)
    c.row8[j] = (unsigned)((c.cell_xy ? c.off0 : 0) + j * a.sy) * 8u;
#pragma unroll
  for (int j = 0; j < P; ++j)
    c.xrow8[j] = (unsigned)(c.off0 + ((exists && !c.ph_y) ? j * a.sy : 0)) * 8u;

  const EoMatrix QF(a.QFc), QB(a.QBc);
  const __attribute__((address_space(4))) double *Dt = (const __attribute__((address_space(4))) double *)a.D;

  const int Zb = zc * a.zlen, Ze = min(Zb + a.zlen, a.nc[2]);
  const int z_first = Zb > 0 ? Zb - 1 : Zb;

  // scratchpad addresses of this lane (doubles): forward planes of this wave, its z columns, its backward planes
  double *const Xl    = lds + lane;
  double *const X_fwd = Xl + (2 * role) * NE * 64;
  double *const X_z   = Xl + (13 * role) * 64;
  const int     nz_mine = role == 0 ? NZC : NZC - 1;
  double *const X_bwd = Xl + (2 * role) * NE * 64;
  const int     kf = 1 + 2 * role, kb = 2 * role; // first forward / backward plane of this wave

  double t0[NZC], carry[NZC];
#pragma unroll
  for (int e = 0; e < NZC; ++e)
    carry[e] = 0.;

  areg_reserve();
  // LDS behind the scratchpad: the scaling table (125) + 5 zeros for the cells that are not computed, the two matrices
  double *const Dl = lds + XLDS;
  double *const Ml = Dl + 136;
  for (int e = threadIdx.x; e < 136; e += 128)
    Dl[e] = e < 125 ? a.D[e] : 0.;
  if (threadIdx.x < 16)
    Ml[threadIdx.x] = a.QFc[threadIdx.x];
  else if (threadIdx.x < 32)
    Ml[threadIdx.x] = a.QBc[threadIdx.x - 16];
  const double *const MF = Ml, *const MB = Ml + 16;
  // this lane's scaling rows: column e of this wave at Dz + Dstep e (cells that are not computed: the zero row)
  const double *const Dz    = c.cell_xy ? Dl + 65 * role : Dl + 125;
  const int           Dstep = c.cell_xy ? 5 : 0;
  __syncthreads();
  // ---- prologue: plane 0 of the first layer (wave 0 transforms it, both pick their columns), requests of the first layer
  {
    const double *s0 = a.src + (long long)(P * z_first) * a.sz;
    plane_request<A_P0>(s0, c);
    plane_request<A_S0>(s0 + (long long)kf * a.sz, c);
    plane_request<A_S1>(s0 + (long long)(kf + 1) * a.sz, c);
    if (z_first >= Zb)
      {
        x_request<A_X0>(a.x + (long long)(P * z_first + kb) * a.sz, c);
        x_request<A_X1>(a.x + (long long)(P * z_first + kb + 1) * a.sz, c);
      }
    vm_wait<0>();
    double v[NE];
    plane_unpack<A_P0>(v);
    const EoVregs qx(MF), qy(MF);
    plane2d<true>(v, qx, qy);
    if (role == 0)
      {
#pragma unroll
        for (int e = 0; e < NE; ++e)
          Xl[e * 64] = v[e];
      }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NZC; ++e)
      t0[e] = e < nz_mine ? X_z[e * 64] : 0.;
    __syncthreads();
  }

  for (int cz = z_first; cz < Ze; ++cz)
    {
      const bool emit = cz >= Zb, more = cz + 1 < Ze;
      STAMP(0);
      // ---- forward x-y transforms of this wave's two planes -> scratchpad
      {
        const EoVregs qx(MF), qy(MF);
        if (cz == z_first || cz == Zb)
          vm_wait<0>(); // (first layer; or the layer below was a halo layer: nothing but requests is outstanding)
        else
          vm_wait<16>(); // younger than the plane requests: the update operands of this layer, the stores of the layer below
        {
          double v[NE];
          plane_unpack<A_S0>(v);
          plane2d<true>(v, qx, qy);
#pragma unroll
          for (int e = 0; e < NE; ++e)
            X_fwd[(0 * NE + e) * 64] = v[e];
        }
        {
          double v[NE];
          plane_unpack<A_S1>(v);
          plane2d<true>(v, qx, qy);
#pragma unroll
          for (int e = 0; e < NE; ++e)
            X_fwd[(1 * NE + e) * 64] = v[e];
        }
        // the next layer's planes: a whole layer ahead
        if (more && !(ABL & 1))
          {
            const double *s1 = a.src + (long long)(P * (cz + 1) + kf) * a.sz;
            plane_request<A_S0>(s1, c);
            plane_request<A_S1>(s1 + a.sz, c);
          }
      }
      // (the z matrices are read ahead of the barrier)
      const EoVregs qfz(MF), qbz(MB);
      __syncthreads();
      STAMP(1);
      // ---- z sweeps of this wave's columns, in place in the scratchpad; the reads run one column ahead
      {
        double n1 = X_z[(0 * NE) * 64], n2 = X_z[(1 * NE) * 64], n3 = X_z[(2 * NE) * 64], n4 = X_z[(3 * NE) * 64];
        double nd[N1];
#pragma unroll
        for (int cc = 0; cc < N1; ++cc)
          nd[cc] = Dz[cc];
#pragma unroll
        for (int e = 0; e < NZC; ++e)
          if (e < NZC - 1 || role == 0)
            {
              double       u0 = t0[e], u1 = n1, u2 = n2, u3 = n3, u4 = n4;
              const double d0 = nd[0], d1 = nd[1], d2 = nd[2], d3 = nd[3], d4 = nd[4];
              if (e + 1 < NZC && (e + 1 < NZC - 1 || role == 0))
                {
                  n1 = X_z[(0 * NE + e + 1) * 64];
                  n2 = X_z[(1 * NE + e + 1) * 64];
                  n3 = X_z[(2 * NE + e + 1) * 64];
                  n4 = X_z[(3 * NE + e + 1) * 64];
#pragma unroll
                  for (int cc = 0; cc < N1; ++cc)
                    nd[cc] = Dz[(e + 1) * Dstep + cc];
                }
              t0[e] = u4;
              line5_eo<true>(u0, u1, u2, u3, u4, qfz);
              u0 *= d0;
              u1 *= d1;
              u2 *= d2;
              u3 *= d3;
              u4 *= d4;
              line5_eo<false>(u0, u1, u2, u3, u4, qbz);
              if (emit)
                {
                  X_z[(0 * NE + e) * 64] = u0 + carry[e];
                  X_z[(1 * NE + e) * 64] = u1;
                  X_z[(2 * NE + e) * 64] = u2;
                  X_z[(3 * NE + e) * 64] = u3;
                }
              carry[e] = u4;
            }
      }
      const EoVregs qx(MB), qy(MB);
      __syncthreads();
      STAMP(2);
      if (emit)
        {
          // ---- backward x-y transforms of this wave's two output planes, merge, update, store
          double w[NE];
#pragma unroll
          for (int e = 0; e < NE; ++e)
            w[e] = X_bwd[(0 * NE + e) * 64];
          if (cz == Zb || !more)
            vm_wait<0>();
          else
            vm_wait<30>(); // younger than the operand requests: the 30 plane requests of this layer
          {
            double xa[P * P];
            x_unpack<A_X0>(xa);
            plane2d<false>(w, qx, qy);
            emit_plane(a, c, P * cz + kb, w, xa);
          }
#pragma unroll
          for (int e = 0; e < NE; ++e)
            w[e] = X_bwd[(1 * NE + e) * 64];
          __syncthreads(); // the scratchpad may take the next layer's planes
          {
            double xa[P * P];
            x_unpack<A_X1>(xa);
            plane2d<false>(w, qx, qy);
            emit_plane(a, c, P * cz + kb + 1, w, xa);
          }
        }
      else
        __syncthreads();
      if (more && !(ABL & 2))
        {
          // the next layer's update operands
          x_request<A_X0>(a.x + (long long)(P * (cz + 1) + kb) * a.sz, c);
          x_request<A_X1>(a.x + (long long)(P * (cz + 1) + kb + 1) * a.sz, c);
        }
      STAMP(3);
    }
  if (threadIdx.x == 0 && blockIdx.x < 4096)
    {
      g_life[blockIdx.x][0] = life_t0;
      g_life[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
      g_hw[blockIdx.x][0]   = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // XCC_ID
      g_hw[blockIdx.x][1]   = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    }
  if (stamp_on && lane == 0)
    {
      for (int i = 0; i < 8; ++i)
        g_stamp[i] = stamp_acc[i];
      g_stamp[7] = Ze - z_first;
    }
  if (Ze == a.nc[2])
    {
      // top plane of the block: the carry columns of both waves, transformed back by wave 0
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NZC; ++e)
        if (e < nz_mine)
          X_z[e * 64] = carry[e];
      __syncthreads();
      if (role == 0)
        {
          double v[NE];
#pragma unroll
          for (int e = 0; e < NE; ++e)
            v[e] = Xl[e * 64];
          x_request<A_X0>(a.x + (long long)(P * a.nc[2]) * a.sz, c);
          vm_wait<0>();
          double xa[P * P];
          x_unpack<A_X0>(xa);
          const EoVregs qx(MB), qy(MB);
          plane2d<false>(v, qx, qy);
          emit_plane(a, c, P * a.nc[2], v, xa);
        }
    }
}

// ---- naive check: one thread per cell, atomics into acc (zeroed), then dst = x + f2 acc
__global__ void naive_kernel(const Args a, double *acc)
{
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long ncell = (long long)a.nc[0] * a.nc[1] * a.nc[2];
  if (t >= ncell)
    return;
  const int cx = t % a.nc[0], cy = (t / a.nc[0]) % a.nc[1], cz = t / ((long long)a.nc[0] * a.nc[1]);
  double    u[125], w[125];
  for (int k = 0; k < 5; ++k)
    for (int j = 0; j < 5; ++j)
      for (int i = 0; i < 5; ++i)
        u[(k * 5 + j) * 5 + i] = a.src[(long long)(P * cz + k) * a.sz + (long long)(P * cy + j) * a.sy + P * cx + i];
  // forward: v[c][b][a] = sum QF[i][a] QF[j][b] QF[k][c] u[k][j][i]
  for (int dir = 0; dir < 3; ++dir)
    {
      const int st = dir == 0 ? 1 : (dir == 1 ? 5 : 25);
      for (int e = 0; e < 125; ++e)
        {
          const int pos = (e / st) % 5, base = e - pos * st;
          double    s   = 0.;
          for (int i = 0; i < 5; ++i)
            s += a.QF[i * 5 + pos] * u[base + i * st];
          w[e] = s;
        }
      for (int e = 0; e < 125; ++e)
        u[e] = w[e];
    }
  for (int c = 0; c < 5; ++c)
    for (int bb = 0; bb < 5; ++bb)
      for (int aa = 0; aa < 5; ++aa)
        u[(c * 5 + bb) * 5 + aa] *= a.D[(bb * 5 + aa) * 5 + c];
  for (int dir = 0; dir < 3; ++dir)
    {
      const int st = dir == 0 ? 1 : (dir == 1 ? 5 : 25);
      for (int e = 0; e < 125; ++e)
        {
          const int pos = (e / st) % 5, base = e - pos * st;
          double    s   = 0.;
          for (int m = 0; m < 5; ++m)
            s += a.QB[pos * 5 + m] * u[base + m * st];
          w[e] = s;
        }
      for (int e = 0; e < 125; ++e)
        u[e] = w[e];
    }
  for (int k = 0; k < 5; ++k)
    for (int j = 0; j < 5; ++j)
      for (int i = 0; i < 5; ++i)
        atomicAdd(&acc[(long long)(P * cz + k) * a.sz + (long long)(P * cy + j) * a.sy + P * cx + i], u[(k * 5 + j) * 5 + i]);
}
__global__ void combine_kernel(double *dst, const double *x, const double *acc, double f2, long long n)
{
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < n)
    dst[t] = x[t] + f2 * acc[t];
}

#define CK(x)                                                                      \
  do                                                                               \
    {                                                                              \
      hipError_t e = (x);                                                          \
      if (e != hipSuccess)                                                         \
        {                                                                          \
          printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
          return 1;                                                                \
        }                                                                          \
    }                                                                              \
  while (0)

int main(int argc, char **argv)
{
  int nc[3] = {80, 80, 80};
  if (argc >= 4)
    for (int d = 0; d < 3; ++d)
      nc[d] = atoi(argv[1 + d]);
  int nzc = argc >= 5 ? atoi(argv[4]) : 0;
  const int nx = P * nc[0] + 1, ny = P * nc[1] + 1, nz = P * nc[2] + 1;
  Args      a;
  a.sy = (nx + 7) / 8 * 8;
  a.sz = (long long)a.sy * ny;
  const long long n = a.sz * nz + 64;
  for (int d = 0; d < 3; ++d)
    a.nc[d] = nc[d];
  a.ntx = nc[0] + 1 <= TX ? 1 : 1 + (nc[0] + 1 - TX + OX - 1) / OX; // (the cells nc own the last DoF column / row)
  a.nty = nc[1] + 1 <= TY ? 1 : 1 + (nc[1] + 1 - TY + OY - 1) / OY;
  int n_cu = 256;
  if (nzc <= 0)
    nzc = std::max(1, (4 * n_cu) / (a.ntx * a.nty)); // (two waves per tile: about two rounds of workgroups)
  nzc    = std::min(nzc, nc[2]);
  a.zlen = (nc[2] + nzc - 1) / nzc;
  a.nzc  = (nc[2] + a.zlen - 1) / a.zlen;
  a.f2   = 0.37;
  // tables: a matrix with even / odd columns, D positive
  std::vector<double> QF(25), QB(25), D(125);
  const double ev[3][3] = {{0.31, -0.52, 0.77}, {0.64, 0.23, -0.41}, {0.83, 0.6, 0.35}}; // [col][i = 0,1,2]
  const double od[2][2] = {{0.45, 0.7}, {-0.58, 0.29}};
  for (int col = 0; col < 3; ++col)
    {
      QF[0 * 5 + col] = QF[4 * 5 + col] = ev[col][0];
      QF[1 * 5 + col] = QF[3 * 5 + col] = ev[col][1];
      QF[2 * 5 + col]                   = ev[col][2];
    }
  for (int col = 0; col < 2; ++col)
    {
      QF[0 * 5 + 3 + col] = od[col][0];
      QF[4 * 5 + 3 + col] = -od[col][0];
      QF[1 * 5 + 3 + col] = od[col][1];
      QF[3 * 5 + 3 + col] = -od[col][1];
      QF[2 * 5 + 3 + col] = 0.;
    }
  const double wsym[5] = {0.5, 1., 1., 1., 0.5};
  for (int i = 0; i < 5; ++i)
    for (int m = 0; m < 5; ++m)
      QB[i * 5 + m] = wsym[i] * QF[i * 5 + m];
  for (int e = 0; e < 125; ++e)
    D[e] = 1.0 / (1.0 + 0.013 * e);
  std::vector<double> hs(n), hx(n);
  unsigned            seed = 12345u;
  for (long long i = 0; i < n; ++i)
    {
      seed  = seed * 1664525u + 1013904223u;
      hs[i] = (double)(seed >> 8) / (1 << 24) - 0.5;
      seed  = seed * 1664525u + 1013904223u;
      hx[i] = (double)(seed >> 8) / (1 << 24) - 0.5;
    }
  std::vector<double> QFc(16, 0.), QBc(16, 0.);
  for (int aa = 0; aa < 3; ++aa)
    for (int i = 0; i < 3; ++i)
      {
        QFc[aa * 3 + i] = QF[i * 5 + aa];
        QBc[aa * 3 + i] = QB[i * 5 + aa];
      }
  for (int aa = 0; aa < 2; ++aa)
    for (int i = 0; i < 2; ++i)
      {
        QFc[9 + aa * 2 + i] = QF[i * 5 + 3 + aa];
        QBc[9 + aa * 2 + i] = QB[i * 5 + 3 + aa];
      }
  double *d_src, *d_x, *d_dst, *d_ref, *d_acc, *d_QF, *d_QB, *d_D, *d_QFc, *d_QBc;
  CK(hipMalloc(&d_QFc, 128));
  CK(hipMalloc(&d_QBc, 128));
  CK(hipMemcpy(d_QFc, QFc.data(), 128, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_QBc, QBc.data(), 128, hipMemcpyHostToDevice));
  a.QFc = d_QFc;
  a.QBc = d_QBc;
  CK(hipMalloc(&d_src, n * 8));
  CK(hipMalloc(&d_x, n * 8));
  CK(hipMalloc(&d_dst, n * 8));
  CK(hipMalloc(&d_ref, n * 8));
  CK(hipMalloc(&d_acc, n * 8));
  CK(hipMalloc(&d_QF, 25 * 8));
  CK(hipMalloc(&d_QB, 25 * 8));
  CK(hipMalloc(&d_D, 125 * 8));
  CK(hipMemcpy(d_src, hs.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, hx.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_QF, QF.data(), 200, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_QB, QB.data(), 200, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_D, D.data(), 1000, hipMemcpyHostToDevice));
  CK(hipMemset(d_dst, 0, n * 8));
  CK(hipMemset(d_acc, 0, n * 8));
  a.src = d_src;
  a.x   = d_x;
  a.dst = d_dst;
  a.QF  = d_QF;
  a.QB  = d_QB;
  a.D   = d_D;
  const int nwaves = a.ntx * a.nty * a.nzc;
  printf("cells %d %d %d  tile %dx%d  tiles %d x %d  z chunks %d of %d layers  waves %d\n", nc[0], nc[1], nc[2], TX, TY, a.ntx, a.nty,
         a.nzc, a.zlen, nwaves);
  // reference
  const long long ncell = (long long)nc[0] * nc[1] * nc[2];
  naive_kernel<<<(unsigned)((ncell + 63) / 64), 64>>>(a, d_acc);
  combine_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d_ref, d_x, d_acc, a.f2, n);
  CK(hipDeviceSynchronize());
  lane_kernel<<<nwaves, 128, (XLDS + 136 + 32) * 8>>>(a);
  CK(hipDeviceSynchronize());
  std::vector<double> r1(n), r2(n);
  CK(hipMemcpy(r1.data(), d_dst, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r2.data(), d_ref, n * 8, hipMemcpyDeviceToHost));
  double num = 0., den = 0., mx = 0.;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i)
        {
          const long long o = k * a.sz + (long long)j * a.sy + i;
          const double    d = r1[o] - r2[o];
          num += d * d;
          den += r2[o] * r2[o];
          mx = std::max(mx, std::fabs(d));
        }
  printf("check: rel l2 diff %.3e  max abs diff %.3e\n", std::sqrt(num / den), mx);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    lane_kernel<<<nwaves, 128, (XLDS + 136 + 32) * 8>>>(a);
  const int reps = 20;
  hipEventRecord(e0);
  for (int rep = 0; rep < reps; ++rep)
    lane_kernel<<<nwaves, 128, (XLDS + 136 + 32) * 8>>>(a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double ndof = (double)nx * ny * nz;
  {
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    const double L = (double)st[7];
    printf("stamps (cycles per layer of one wave, %d layers): emit-tail->top %.0f  fwd %.0f  z-stage %.0f  bwd+emit %.0f\n", (int)st[7], st[0] / L, st[1] / L,
           st[2] / L, st[3] / L);
  }
  {
    static unsigned long long lf[4096][2];
    hipMemcpyFromSymbol(lf, HIP_SYMBOL(g_life), sizeof(lf));
    const int nb = nwaves < 4096 ? nwaves : 4096;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nb; ++i)
      {
        t0 = lf[i][0] < t0 ? lf[i][0] : t0;
        t1 = lf[i][1] > t1 ? lf[i][1] : t1;
      }
    double avg_life = 0., avg_start = 0., max_start = 0.;
    for (int i = 0; i < nb; ++i)
      {
        avg_life += (double)(lf[i][1] - lf[i][0]);
        avg_start += (double)(lf[i][0] - t0);
        max_start = std::max(max_start, (double)(lf[i][0] - t0));
      }
    {
      std::vector<double> lv(nb);
      for (int i = 0; i < nb; ++i)
        lv[i] = (double)(lf[i][1] - lf[i][0]) / 100.;
      std::vector<double> sl(lv);
      std::sort(sl.begin(), sl.end());
      printf("life percentiles us: min %.1f 10%% %.1f 50%% %.1f 90%% %.1f 99%% %.1f max %.1f\n", sl[0], sl[nb / 10], sl[nb / 2], sl[nb * 9 / 10], sl[nb * 99 / 100], sl[nb - 1]);
      double xs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int    xn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < nb; ++i)
        {
          xs[i & 7] += lv[i];
          xn[i & 7]++;
        }
      printf("average life by block %% 8:");
      for (int i = 0; i < 8; ++i)
        printf(" %.1f", xs[i] / xn[i]);
      printf("\n");
      {
        static unsigned hw[4096][2];
        hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_hw), sizeof(hw));
        double xa[16] = {}, sa[4] = {}, ca[16] = {};
        int    xc[16] = {}, sc[4] = {}, cc[16] = {};
        for (int i = 0; i < nb; ++i)
          {
            const unsigned xcc = hw[i][0] & 15, simd = (hw[i][1] >> 4) & 3, cu = (hw[i][1] >> 8) & 15, sh = (hw[i][1] >> 12) & 1, se = (hw[i][1] >> 13) & 7;
            xa[xcc] += lv[i];
            xc[xcc]++;
            sa[simd] += lv[i];
            sc[simd]++;
            ca[cu] += lv[i];
            cc[cu]++;
            if (i < 6)
              printf("  block %d: xcc %u se %u sh %u cu %u simd %u life %.1f\n", i, xcc, se, sh, cu, simd, lv[i]);
          }
        {
          // per compute unit: key = xcc, se, sh, cu
          double cu_sum[8 * 8 * 2 * 16] = {};
          double cu_min[8 * 8 * 2 * 16], cu_max[8 * 8 * 2 * 16] = {};
          int    cu_n[8 * 8 * 2 * 16] = {};
          for (int i = 0; i < 8 * 8 * 2 * 16; ++i)
            cu_min[i] = 1e30;
          for (int i = 0; i < nb; ++i)
            {
              const unsigned xcc = hw[i][0] & 7, cu = (hw[i][1] >> 8) & 15, sh = (hw[i][1] >> 12) & 1, se = (hw[i][1] >> 13) & 7;
              const int      key = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
              cu_sum[key] += lv[i];
              cu_n[key]++;
              cu_min[key] = std::min(cu_min[key], lv[i]);
              cu_max[key] = std::max(cu_max[key], lv[i]);
            }
          int    ncu = 0, hist[8] = {};
          double spread = 0.;
          std::vector<double> cavg;
          for (int k = 0; k < 8 * 8 * 2 * 16; ++k)
            if (cu_n[k])
              {
                ++ncu;
                hist[std::min(cu_n[k], 7)]++;
                spread += cu_max[k] - cu_min[k];
                cavg.push_back(cu_sum[k] / cu_n[k]);
              }
          std::sort(cavg.begin(), cavg.end());
          printf("compute units seen %d; waves per CU histogram: 1:%d 2:%d 3:%d 4:%d 5+:%d; mean (max-min) life inside a CU %.1f us; CU average life min %.1f median %.1f max %.1f\n", ncu,
                 hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], spread / ncu, cavg.front(), cavg[cavg.size() / 2], cavg.back());
        }
        if (const char *dump = getenv("LANE_DUMP"))
          {
            FILE *f = fopen(dump, "w");
            for (int i = 0; i < nb; ++i)
              fprintf(f, "%d %u %u %u %u %u %llu %llu\n", i, hw[i][0] & 7, (hw[i][1] >> 13) & 7, (hw[i][1] >> 12) & 1, (hw[i][1] >> 8) & 15, (hw[i][1] >> 4) & 3,
                      lf[i][0] - t0, lf[i][1] - t0);
            fclose(f);
          }
        printf("average life by XCC:");
        for (int i = 0; i < 8; ++i)
          printf(" %.1f(%d)", xc[i] ? xa[i] / xc[i] : 0., xc[i]);
        printf("\naverage life by SIMD:");
        for (int i = 0; i < 4; ++i)
          printf(" %.1f(%d)", sc[i] ? sa[i] / sc[i] : 0., sc[i]);
        printf("\naverage life by CU id:");
        for (int i = 0; i < 16; ++i)
          printf(" %.1f(%d)", cc[i] ? ca[i] / cc[i] : 0., cc[i]);
        printf("\n");
      }
      // by z chunk of the work index
      const int per = a.ntx * a.nty;
      std::vector<double> zs(a.nzc, 0.);
      for (int b = 0; b < nb; ++b)
        {
          const int xcd = b & 7, idx = b >> 3;
          const int wi  = xcd * (nwaves >> 3) + std::min(xcd, nwaves & 7) + idx;
          zs[wi / per] += lv[b] / per;
        }
      printf("average life by z chunk:");
      for (int z = 0; z < a.nzc; ++z)
        printf(" %.1f", zs[z]);
      printf("\n");
    }
    printf("block lives (100 MHz ticks -> us): span %.1f us, average life %.1f us, average start %.1f us, last start %.1f us\n", (t1 - t0) / 100., avg_life / nb / 100.,
           avg_start / nb / 100., max_start / 100.);
  }
  printf("lane sweep: %.4f ms  %.3e DoF/s  %.1f GB/s (24 B/DoF) = %.3f of 8 TB/s\n", ms, ndof / ms * 1e3, 24. * ndof / ms / 1e6,
         24. * ndof / ms / 1e6 / 8000.);
  return 0;
}
