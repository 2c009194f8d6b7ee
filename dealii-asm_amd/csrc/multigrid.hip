// The smoother's callers: level hierarchy, transfer operators, multigrid V-cycle, coarse
// solver and the outer Krylov solvers, on top of the library's own C ABI (every level operator /
// smoother is an asm_ctx; nothing here reaches into a context).
//
// Reference:
//   include/multigrid.h:260-472                           PreconditionerGMG::do_update / vmult
//   include/precondition.h:82-186                         MyMultigrid (smoother / coarse solver factories)
//   element_centered_preconditioners_01.cc:540-740        level construction
//   element_centered_preconditioners_01.cc:108-263        solve()
//   transfer_01.likwid.cc:21-137                          MGTwoLevelTransfer between two operators
// and, inside deal.II, Multigrid::level_v_step, MGTransferGlobalCoarsening, SolverCG, SolverGMRES.
//
// Transfers: both levels are tensor-product spaces on nested structured meshes, so the prolongation is
// the Kronecker product of three 1-D nodal interpolation matrices (rows: fine nodes, columns: the
// coarse basis functions of the coarse cell that contains the node).  Prolongation gathers per fine
// DoF, restriction (the transpose) gathers per coarse DoF over the fine nodes in the support of its
// basis function: no atomics, every entry written once.
#include "../../include/dealii_asm_amd.h"
#include "numerics.h"

#include <hip/hip_runtime.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#  error "written for gfx950 (wave64): the reductions of dense_apply_kernel assume 64 lanes"
#endif

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace asmamd;

namespace
{
  thread_local std::string g_mg_create_error;

  constexpr int MAX_N1C = 8;  // coarse nodes per cell and direction (degree <= 7)
  constexpr int MAX_SUP = 15; // fine nodes in the 1-D support of a coarse basis function (2 * 7 + 1)

  // 1-D tables of one direction (device pointers)
  struct Transfer1D
  {
    // prolongation: fine logical index I -> first coarse logical index, count and weights (zeros at both ends of
    // the row trimmed: a fine node that coincides with a coarse node has one entry)
    const int *   p_start;
    const int *   p_count;
    const double *p_val; // [n_fine][MAX_N1C]
    // restriction: coarse logical index J -> first fine logical index, count and weights
    const int *   r_start;
    const int *   r_count;
    const double *r_val; // [n_coarse][MAX_SUP]
    int           n_fine, n_coarse, n1c;
  };
  struct BoxLayout
  {
    int       begin[3]; // first stored logical coordinate
    int       size[3];  // owned planes [0, size)
    long long stride[3];
  };
  struct TransferArgs
  {
    Transfer1D d[3];
    BoxLayout  fine, coarse;
  };

  __device__ __forceinline__ long long
  box_offset(const BoxLayout &b, const int I, const int J, const int K)
  {
    return (I - b.begin[0]) * b.stride[0] + (J - b.begin[1]) * b.stride[1] + (K - b.begin[2]) * b.stride[2];
  }

  // One direction of a transfer (the Kronecker factors applied one after the other through two intermediate
  // boxes, each pass at streaming rate): out[I, J, K] (+)= sum_a val[idx][a] in[.. start[idx] + a ..], idx the
  // output coordinate along dir.
  struct PassArgs
  {
    const int *   start;
    const int *   count;
    const double *val;
    int           val_stride, dir;
    BoxLayout     in, out; // out.size = loop bounds
  };

  template <bool ADD>
  __global__ void __launch_bounds__(256)
  transfer_pass_kernel(const PassArgs a, double *__restrict__ out, const double *__restrict__ in)
  {
    const long long n = (long long)a.out.size[0] * a.out.size[1] * a.out.size[2];
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
      {
        // (32-bit index arithmetic: the boxes of a level have fewer than 2^31 entries, checked by the launcher; 64-bit
        //  divisions cost several hundred instructions per entry)
        const unsigned  eu = (unsigned)e, s0 = (unsigned)a.out.size[0], s01 = s0 * (unsigned)a.out.size[1];
        const unsigned  ck = eu / s01, rem = eu - ck * s01, cj = rem / s0;
        int             c[3] = {(int)(rem - cj * s0), (int)cj, (int)ck};
        const long long o  = box_offset(a.out, c[0], c[1], c[2]);
        const int       id = c[a.dir];
        const int       cnt = a.count[id];
        const double *  w   = a.val + (long long)id * a.val_stride;
        c[a.dir]            = a.start[id];
        const double *  row = in + box_offset(a.in, c[0], c[1], c[2]);
        const long long st  = a.in.stride[a.dir];
        double          sum = 0.;
        for (int k = 0; k < cnt; ++k)
          sum = fma(w[k], row[k * st], sum);
        out[o] = ADD ? out[o] + sum : sum;
      }
  }

  // the z direction: one grid row (blockIdx.y) per output plane, so that the plane's table entries (start, count,
  // weights) are block-uniform and come through scalar loads, and the loop has a uniform trip count.  A thread takes
  // PU outputs of the plane (a quarter of the plane apart) and issues their loads together: with one output per thread
  // a thread is a single round trip to memory and the passes ran at 2-3.7 TB/s (profiles/r03_transfer_passes.txt)
  constexpr int PU = 4;
  template <bool ADD>
  __global__ void __launch_bounds__(256)
  transfer_pass_z_kernel(const PassArgs a, double *__restrict__ out, const double *__restrict__ in)
  {
    const unsigned  s0 = (unsigned)a.out.size[0], n2 = s0 * (unsigned)a.out.size[1], span = gridDim.x * 256u;
    const int       ck = (int)blockIdx.y, cnt = a.count[ck];
    const double *  w  = a.val + (long long)ck * a.val_stride;
    const long long st = a.in.stride[2];
    const double *  row[PU];
    long long       o[PU];
    double          sum[PU], prev[PU];
    bool            on[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u)
      {
        const unsigned e2 = blockIdx.x * 256u + threadIdx.x + u * span;
        on[u]             = e2 < n2;
        const unsigned ee = on[u] ? e2 : 0u, cj = ee / s0;
        const int      ci = (int)(ee - cj * s0);
        o[u]              = box_offset(a.out, ci, (int)cj, ck);
        row[u]            = in + box_offset(a.in, ci, (int)cj, a.start[ck]);
        sum[u]            = 0.;
        prev[u]           = (ADD && on[u]) ? out[o[u]] : 0.;
      }
    for (int k = 0; k < cnt; ++k)
      {
        double v[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u)
          v[u] = row[u][k * st];
#pragma unroll
        for (int u = 0; u < PU; ++u)
          sum[u] = fma(w[k], v[u], sum[u]);
      }
#pragma unroll
    for (int u = 0; u < PU; ++u)
      if (on[u])
        out[o[u]] = ADD ? prev[u] + sum[u] : sum[u];
  }

  // the y direction in the same way: one grid row per output ROW index j, the threads of a block run over the (x, z) pairs
  template <bool ADD>
  __global__ void __launch_bounds__(256)
  transfer_pass_y_kernel(const PassArgs a, double *__restrict__ out, const double *__restrict__ in)
  {
    const unsigned  s0 = (unsigned)a.out.size[0], n2 = s0 * (unsigned)a.out.size[2], span = gridDim.x * 256u;
    const int       cj = (int)blockIdx.y, cnt = a.count[cj];
    const double *  w  = a.val + (long long)cj * a.val_stride;
    const long long st = a.in.stride[1];
    const double *  row[PU];
    long long       o[PU];
    double          sum[PU], prev[PU];
    bool            on[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u)
      {
        const unsigned e2 = blockIdx.x * 256u + threadIdx.x + u * span;
        on[u]             = e2 < n2;
        const unsigned ee = on[u] ? e2 : 0u, ck = ee / s0;
        const int      ci = (int)(ee - ck * s0);
        o[u]              = box_offset(a.out, ci, cj, (int)ck);
        row[u]            = in + box_offset(a.in, ci, a.start[cj], (int)ck);
        sum[u]            = 0.;
        prev[u]           = (ADD && on[u]) ? out[o[u]] : 0.;
      }
    for (int k = 0; k < cnt; ++k)
      {
        double v[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u)
          v[u] = row[u][k * st];
#pragma unroll
        for (int u = 0; u < PU; ++u)
          sum[u] = fma(w[k], v[u], sum[u]);
      }
#pragma unroll
    for (int u = 0; u < PU; ++u)
      if (on[u])
        out[o[u]] = ADD ? prev[u] + sum[u] : sum[u];
  }

  // dst_fine += P src_coarse: one thread per fine DoF
  __global__ void __launch_bounds__(256)
  prolongate_add_kernel(const TransferArgs t, double *__restrict__ dst, const double *__restrict__ src)
  {
    const long long n = (long long)t.fine.size[0] * t.fine.size[1] * t.fine.size[2];
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
      {
        const int I = (int)(e % t.fine.size[0]), J = (int)((e / t.fine.size[0]) % t.fine.size[1]),
                  K = (int)(e / ((long long)t.fine.size[0] * t.fine.size[1]));
        const int     cx = t.d[0].p_start[I], cy = t.d[1].p_start[J], cz = t.d[2].p_start[K];
        const int     nx = t.d[0].p_count[I], ny = t.d[1].p_count[J], nz = t.d[2].p_count[K];
        const double *wx = t.d[0].p_val + (long long)I * MAX_N1C, *wy = t.d[1].p_val + (long long)J * MAX_N1C,
                     *wz = t.d[2].p_val + (long long)K * MAX_N1C;
        double sum = 0.;
        for (int c = 0; c < nz; ++c)
          {
            double sy = 0.;
            for (int b = 0; b < ny; ++b)
              {
                const double *row = src + box_offset(t.coarse, cx, cy + b, cz + c);
                double        sx  = 0.;
                for (int a = 0; a < nx; ++a)
                  sx = fma(wx[a], row[a * t.coarse.stride[0]], sx);
                sy = fma(wy[b], sx, sy);
              }
            sum = fma(wz[c], sy, sum);
          }
        dst[box_offset(t.fine, I, J, K)] += sum;
      }
  }

  // dst_coarse = P^T src_fine: one thread per coarse DoF
  __global__ void __launch_bounds__(256)
  restrict_kernel(const TransferArgs t, double *__restrict__ dst, const double *__restrict__ src)
  {
    const long long n = (long long)t.coarse.size[0] * t.coarse.size[1] * t.coarse.size[2];
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256)
      {
        const int I = (int)(e % t.coarse.size[0]), J = (int)((e / t.coarse.size[0]) % t.coarse.size[1]),
                  K = (int)(e / ((long long)t.coarse.size[0] * t.coarse.size[1]));
        const int     fx = t.d[0].r_start[I], fy = t.d[1].r_start[J], fz = t.d[2].r_start[K];
        const int     nx = t.d[0].r_count[I], ny = t.d[1].r_count[J], nz = t.d[2].r_count[K];
        const double *wx = t.d[0].r_val + (long long)I * MAX_SUP, *wy = t.d[1].r_val + (long long)J * MAX_SUP,
                     *wz = t.d[2].r_val + (long long)K * MAX_SUP;
        double sum = 0.;
        for (int c = 0; c < nz; ++c)
          {
            double sy = 0.;
            for (int b = 0; b < ny; ++b)
              {
                const double *row = src + box_offset(t.fine, fx, fy + b, fz + c);
                double        sx  = 0.;
                for (int a = 0; a < nx; ++a)
                  sx = fma(wx[a], row[a * t.fine.stride[0]], sx);
                sy = fma(wy[b], sx, sy);
              }
            sum = fma(wz[c], sy, sum);
          }
        dst[box_offset(t.coarse, I, J, K)] = sum;
      }
  }

  // x = Ainv b on the owned box (the dense inverse of the coarsest level's operator): one workgroup per row
  __global__ void __launch_bounds__(256)
  dense_apply_kernel(const double *__restrict__ Ainv, const int n, const BoxLayout box, double *__restrict__ x, const double *__restrict__ b)
  {
    __shared__ double part[4];
    const int         r   = blockIdx.x;
    double            sum = 0.;
    for (int c = threadIdx.x; c < n; c += 256)
      {
        const int I = c % box.size[0], J = (c / box.size[0]) % box.size[1], K = c / (box.size[0] * box.size[1]);
        sum         = fma(Ainv[(long long)r * n + c], b[box_offset(box, I, J, K)], sum);
      }
    for (int o = 32; o > 0; o >>= 1)
      sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0)
      part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
      {
        const int I = r % box.size[0], J = (r / box.size[0]) % box.size[1], K = r / (box.size[0] * box.size[1]);
        x[box_offset(box, I, J, K)] = (part[0] + part[1]) + (part[2] + part[3]);
      }
  }

  struct Level
  {
    asm_params params;
    asm_ctx *  ctx = nullptr;
    asm_layout layout;
    double *   x = nullptr, *b = nullptr, *r = nullptr; // solution, defect, work vector
    // transfer to the next coarser level (levels >= 1)
    TransferArgs        transfer;
    // intermediate boxes of the direction-by-direction transfer: [fine x][fine y][coarse z], [fine x][coarse y][coarse z]
    double *            pass_xy = nullptr, *pass_x = nullptr;
    std::vector<void *> allocs;
    // Multi-rank hierarchies (include/multigrid.h:260-465 runs every level on the same partition): a level is block-
    // decomposed by the fine level's process grid as long as the blocks stay nested (every rank's cells of the finer
    // level are the children of its cells here); the levels below are held whole by every rank ("replicated": the
    // same computation everywhere, as the reference's coarse levels shrink to a sub-communicator,
    // element_centered_preconditioners_01.cc:592-633).
    bool distributed = false;
  };
} // namespace

// context.hip: pool of solver work vectors of a context (not part of the public interface)
extern "C" int  asm_work_vector_acquire(asm_ctx *ctx, double **v);
extern "C" void asm_work_vector_release(asm_ctx *ctx, double *v);
// context.hip: ghost exchange, all-reduce and communicator sharing of level contexts (multi-rank hierarchies)
extern "C" int asm_internal_exchange(asm_ctx *ctx, double *v, int add);
extern "C" int asm_internal_zero_ghosts(asm_ctx *ctx, double *v);
extern "C" int asm_internal_allreduce(asm_ctx *ctx, double *v, long long n);
extern "C" int asm_internal_comm_share(asm_ctx *dst, asm_ctx *src);
extern "C" void *asm_internal_swap_stream(asm_ctx *ctx, void *stream);
extern "C" void asm_internal_cheb_state(const asm_ctx *ctx, double out[4]);

struct asm_mg
{
  asm_mg_params      mp;
  std::vector<Level> levels;
  hipStream_t        stream = nullptr;
  int                device = 0;
  // coarse CG work vectors (level 0)
  double *    cg_p = nullptr, *cg_z = nullptr, *cg_q = nullptr;
  // ASM_MG_COARSE_DIRECT: dense inverse of the level-0 operator, [n][n] over the owned DoFs
  double *    coarse_inverse = nullptr;
  int         coarse_n       = 0;
  // the last level of degree 1 (element_centered_preconditioners_01.cc:677-685): top of the intermediate multigrid
  int         intermediate_level = 0;
  double *    rc_r = nullptr, *rc_d = nullptr; // "n coarse cycles" > 1: residual and correction on that level
  long long   coarse_not_converged = 0;        // coarse CG solves that hit their iteration limit
  // hipGraph of the legs at and below graph_level (small levels: some sixty launches of 5-10 us each, bound by the launch
  // rate): captured on the second V-cycle, replayed while the levels' Chebyshev parameters stand.  -1: none
  int                 graph_level = -1;
  int                 graph_state = 0; // 0: next V-cycle runs eagerly (lazy kernel set-up), 1: capture / replay, -1: off
  hipStream_t         graph_stream = nullptr;
  hipGraphExec_t      graph_exec   = nullptr;
  std::vector<double> graph_signature;
  long long           graph_replays = 0;
  std::string error;
};

namespace
{
  struct DeviceGuard
  {
    int  prev = -1;
    bool switched = false;
    explicit DeviceGuard(const int device)
    {
      if (hipGetDevice(&prev) == hipSuccess && prev != device)
        switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard()
    {
      if (switched)
        (void)hipSetDevice(prev);
    }
  };

#define MG_TRY(expr)                                                       \
  do                                                                       \
    {                                                                      \
      const int e_ = (expr);                                               \
      if (e_ != ASM_OK)                                                    \
        return e_;                                                         \
    }                                                                      \
  while (0)
// a call into a level context: on failure carry its message
#define MG_CTX(mg_, ctx_, expr)                                            \
  do                                                                       \
    {                                                                      \
      const int e_ = (expr);                                               \
      if (e_ != ASM_OK)                                                    \
        {                                                                  \
          (mg_)->error = asm_last_error(ctx_);                             \
          return e_;                                                       \
        }                                                                  \
    }                                                                      \
  while (0)

  template <typename T>
  bool
  upload(Level &lv, const std::vector<T> &h, const T **d)
  {
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess)
      return false;
    lv.allocs.push_back(p);
    if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return false;
    *d = (const T *)p;
    return true;
  }

  std::vector<int>
  polynomial_sequence(const int p, const int kind)
  {
    // MGTransferGlobalCoarseningTools::create_polynomial_coarsening_sequence (deal.II): ascending
    std::vector<int> seq = {p};
    while (seq.back() > 1)
      {
        const int q = seq.back();
        seq.push_back(kind == ASM_MG_BISECT ? std::max(q / 2, 1) : (kind == ASM_MG_DECREASE_BY_ONE ? q - 1 : 1));
      }
    std::reverse(seq.begin(), seq.end());
    return seq;
  }

  struct LevelSpec
  {
    int nc[3], p;
    bool operator==(const LevelSpec &o) const { return nc[0] == o.nc[0] && nc[1] == o.nc[1] && nc[2] == o.nc[2] && p == o.p; }
  };

  // element_centered_preconditioners_01.cc:640-672.  The 'ph' / 'hp' lists name the joint level twice; both copies are
  // levels (identity transfer, a smoother each) unless mp.keep_duplicate_level == 0
  std::vector<LevelSpec>
  level_sequence(const asm_params &fine, const asm_mg_params &mp)
  {
    const std::vector<int>          degs = polynomial_sequence(fine.degree, mp.p_sequence);
    std::vector<std::array<int, 3>> meshes = {{fine.n_cells[0], fine.n_cells[1], fine.n_cells[2]}};
    // create_geometric_coarsening_sequence on a structured box: halve while every direction is even
    while (meshes.back()[0] % 2 == 0 && meshes.back()[1] % 2 == 0 && meshes.back()[2] % 2 == 0)
      meshes.push_back({meshes.back()[0] / 2, meshes.back()[1] / 2, meshes.back()[2] / 2});
    std::reverse(meshes.begin(), meshes.end());
    std::vector<LevelSpec> lv;
    const auto             add = [&](const std::array<int, 3> &m, const int p) {
      const LevelSpec s{{m[0], m[1], m[2]}, p};
      if (mp.keep_duplicate_level != 0 || lv.empty() || !(lv.back() == s))
        lv.push_back(s);
    };
    switch (mp.mg_type)
      {
        case ASM_MG_H:
          for (const auto &m : meshes)
            add(m, degs.back());
          break;
        case ASM_MG_P:
          for (const int p : degs)
            add(meshes.back(), p);
          break;
        case ASM_MG_HP:
          for (const int p : degs)
            add(meshes.front(), p);
          for (const auto &m : meshes)
            add(m, degs.back());
          break;
        default: // ASM_MG_PH
          for (const auto &m : meshes)
            add(m, degs.front());
          for (const int p : degs)
            add(meshes.back(), p);
      }
    return lv;
  }

  // 1-D interpolation tables between two levels in direction d (host), then uploaded
  // The window of a level pair in direction d (one rank's part of the two levels; whole levels on one rank):
  // fine rows [f_begin, f_begin + f_size) (global DoF coordinates: the rank's owned fine planes) and coarse
  // rows [c_begin, c_begin + c_size) (the coarse planes the rank computes in a restriction: its owned ones and, below an
  // interface, the plane it shares with the next rank, whose owner adds the two parts - compress(add)); coarse planes
  // [c_store_begin, c_store_end) are stored on the rank (read by the prolongation: owned + ghost planes).
  struct Window1D
  {
    int f_begin, f_size, c_begin, c_size, c_store_begin, c_store_end;
  };

  bool
  build_transfer_1d(Level &lv, const asm_params &pf, const asm_params &pc, const int d, const Window1D &w, Transfer1D &out,
                    std::string &error)
  {
    const int ncf = pf.n_cells[d], ncc = pc.n_cells[d], degf = pf.degree, degc = pc.degree;
    int       ratio = 1;
    if (ncf != ncc)
      {
        if (ncf != 2 * ncc || degf != degc)
          {
            error = "transfer: levels are neither p- nor h-nested";
            return false;
          }
        ratio = 2;
      }
    if (pf.boundary[d][0] == ASM_BC_PERIODIC)
      {
        error = "multigrid transfers support non-periodic directions only";
        return false;
      }
    const int                 nf = ncf * degf + 1, nc = ncc * degc + 1, n1f = degf + 1, n1c = degc + 1;
    const std::vector<double> nodes_f = gauss_lobatto(n1f), nodes_c = gauss_lobatto(n1c);
    const auto constrained = [&](const asm_params &p, const int n, const int I) {
      return (I == 0 && p.boundary[d][0] == ASM_BC_DIRICHLET) || (I == n - 1 && p.boundary[d][1] == ASM_BC_DIRICHLET);
    };
    // dense 1-D matrix P[I][J] (small: built once per level pair)
    std::vector<double> P((size_t)nf * nc, 0.);
    std::vector<int>    p_start(nf, 0);
    for (int cf = 0; cf < ncf; ++cf)
      {
        const int           cc = cf / ratio, s = cf % ratio;
        std::vector<double> xi(n1f), V, D;
        for (int i = 0; i < n1f; ++i)
          xi[i] = (s + nodes_f[i]) / ratio;
        lagrange_eval(nodes_c, xi, V, D);
        // shared fine nodes get the same row from both cells; take the lower cell for all but the last node
        for (int i = 0; i < n1f; ++i)
          {
            const int I = degf * cf + i;
            if (i == n1f - 1 && cf + 1 < ncf)
              continue;
            p_start[I] = degc * cc;
            for (int j = 0; j < n1c; ++j)
              P[(size_t)I * nc + degc * cc + j] = V[i * n1c + j];
          }
      }
    for (int I = 0; I < nf; ++I)
      for (int J = 0; J < nc; ++J)
        if (constrained(pf, nf, I) || constrained(pc, nc, J))
          P[(size_t)I * nc + J] = 0.;
    // ---- tables of the window, in local logical coordinates (fine: relative to f_begin, coarse: relative to c_begin)
    std::vector<double> p_val((size_t)w.f_size * MAX_N1C, 0.);
    std::vector<int>    p_count(w.f_size, 0), p_start_w(w.f_size, 0);
    for (int I = w.f_begin; I < w.f_begin + w.f_size; ++I)
      {
        int lo = n1c, hi = -1;
        for (int j = 0; j < n1c; ++j)
          if (P[(size_t)I * nc + p_start[I] + j] != 0.)
            {
              lo = std::min(lo, j);
              hi = std::max(hi, j);
            }
        if (hi < lo)
          continue; // constrained row
        const int st = p_start[I] + lo, li = I - w.f_begin;
        if (st < w.c_store_begin || st + (hi - lo) >= w.c_store_end)
          {
            error = "transfer: a fine row of this rank needs coarse values it does not store (the levels' blocks are not nested)";
            return false;
          }
        p_start_w[li] = st - w.c_begin;
        p_count[li]   = hi - lo + 1;
        for (int j = 0; j < p_count[li]; ++j)
          p_val[(size_t)li * MAX_N1C + j] = P[(size_t)I * nc + st + j];
      }
    // restriction: the fine rows of this rank only (every fine DoF is counted by its owner)
    std::vector<int>    r_start(w.c_size, 0), r_count(w.c_size, 0);
    std::vector<double> r_val((size_t)w.c_size * MAX_SUP, 0.);
    for (int J = w.c_begin; J < w.c_begin + w.c_size; ++J)
      {
        int lo = nf, hi = -1;
        for (int I = w.f_begin; I < w.f_begin + w.f_size; ++I)
          if (P[(size_t)I * nc + J] != 0.)
            {
              lo = std::min(lo, I);
              hi = std::max(hi, I);
            }
        if (hi < lo)
          continue;
        if (hi - lo + 1 > MAX_SUP)
          {
            error = "transfer: support of a coarse basis function exceeds the table width";
            return false;
          }
        const int lj = J - w.c_begin;
        r_start[lj]  = lo - w.f_begin;
        r_count[lj]  = hi - lo + 1;
        for (int I = lo; I <= hi; ++I)
          r_val[(size_t)lj * MAX_SUP + I - lo] = P[(size_t)I * nc + J];
      }
    out.n_fine   = w.f_size;
    out.n_coarse = w.c_size;
    out.n1c      = n1c;
    if (!upload(lv, p_start_w, &out.p_start) || !upload(lv, p_count, &out.p_count) || !upload(lv, p_val, &out.p_val) || !upload(lv, r_start, &out.r_start) ||
        !upload(lv, r_count, &out.r_count) || !upload(lv, r_val, &out.r_val))
      {
        error = "transfer: device allocation failed";
        return false;
      }
    return true;
  }

  BoxLayout
  box_of(const asm_layout &l)
  {
    BoxLayout b;
    for (int d = 0; d < 3; ++d)
      {
        b.begin[d]  = l.store_begin[d];
        b.size[d]   = l.owned_size[d];
        b.stride[d] = l.stride[d];
      }
    return b;
  }

  int
  launch_blocks(const long long n)
  {
    return (int)std::min<long long>(std::max<long long>((n + 255) / 256, 1), 1 << 16);
  }

  BoxLayout
  dense_box(const int nx, const int ny, const int nz)
  {
    BoxLayout b;
    b.begin[0] = b.begin[1] = b.begin[2] = 0;
    b.size[0]                            = nx;
    b.size[1]                            = ny;
    b.size[2]                            = nz;
    b.stride[0]                          = 1;
    b.stride[1]                          = nx;
    b.stride[2]                          = (long long)nx * ny;
    return b;
  }

  PassArgs
  prolongation_pass(const TransferArgs &t, const int dir, const BoxLayout &in, const BoxLayout &out)
  {
    return PassArgs{t.d[dir].p_start, t.d[dir].p_count, t.d[dir].p_val, MAX_N1C, dir, in, out};
  }

  PassArgs
  restriction_pass(const TransferArgs &t, const int dir, const BoxLayout &in, const BoxLayout &out)
  {
    return PassArgs{t.d[dir].r_start, t.d[dir].r_count, t.d[dir].r_val, MAX_SUP, dir, in, out};
  }

  template <bool ADD>
  void
  launch_pass(asm_mg *mg, const PassArgs &a, double *out, const double *in)
  {
    const long long n = (long long)a.out.size[0] * a.out.size[1] * a.out.size[2];
    if (a.dir == 2 && a.out.size[2] <= 65535)
      {
        const long long plane = (long long)a.out.size[0] * a.out.size[1];
        hipLaunchKernelGGL((transfer_pass_z_kernel<ADD>), dim3((unsigned)((plane + 256 * PU - 1) / (256 * PU)), a.out.size[2]), dim3(256), 0, mg->stream, a, out, in);
        return;
      }
    if (a.dir == 1 && a.out.size[1] <= 65535)
      {
        const long long xz = (long long)a.out.size[0] * a.out.size[2];
        hipLaunchKernelGGL((transfer_pass_y_kernel<ADD>), dim3((unsigned)((xz + 256 * PU - 1) / (256 * PU)), a.out.size[1]), dim3(256), 0, mg->stream, a, out, in);
        return;
      }
    hipLaunchKernelGGL((transfer_pass_kernel<ADD>), dim3(launch_blocks(n)), dim3(256), 0, mg->stream, a, out, in);
  }

  int
  check_launch(asm_mg *mg, const char *what)
  {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess)
      return ASM_OK;
    mg->error = std::string("kernel launch failed in ") + what + ": " + hipGetErrorString(e);
    return ASM_ERR_DEVICE;
  }

  // ---- Krylov solvers on one context --------------------------------------------------------------
  // Vectors the solvers see have zeros on constrained DoFs (the right-hand side is cleaned on entry), so the
  // operator's zero rows there and the identity rows of deal.II's operator give the same iterates.
  struct Preconditioner
  {
    int      kind;
    asm_ctx *ctx;
    asm_mg * mg;
  };

  int
  apply_preconditioner(const Preconditioner &pc, std::string &error, double *dst, const double *src)
  {
    int err = ASM_OK;
    switch (pc.kind)
      {
        case ASM_SP_IDENTITY: err = asm_vec_copy(pc.ctx, dst, src); break;
        case ASM_SP_INNER: err = asm_precond_vmult(pc.ctx, dst, src); break;
        case ASM_SP_CHEBYSHEV: err = asm_cheb_vmult(pc.ctx, dst, src); break;
        case ASM_SP_MULTIGRID:
          err = asm_mg_vmult(pc.mg, dst, src);
          if (err)
            error = asm_mg_last_error(pc.mg);
          return err;
        default: error = "unknown system preconditioner"; return ASM_ERR_INVALID_VALUE;
      }
    if (err)
      error = asm_last_error(pc.ctx);
    return err;
  }

#define KS_CTX(expr)                       \
  do                                       \
    {                                      \
      const int e_ = (expr);               \
      if (e_ != ASM_OK)                    \
        {                                  \
          error = asm_last_error(ctx);     \
          return e_;                       \
        }                                  \
    }                                      \
  while (0)
#define KS_TRY(expr)                       \
  do                                       \
    {                                      \
      const int e_ = (expr);               \
      if (e_ != ASM_OK)                    \
        return e_;                         \
    }                                      \
  while (0)

  struct WorkVectors
  {
    asm_ctx *             ctx;
    std::vector<double *> v;
    explicit WorkVectors(asm_ctx *c)
      : ctx(c)
    {}
    ~WorkVectors()
    {
      for (double *p : v)
        asm_work_vector_release(ctx, p);
    }
    double *get()
    {
      double *p = nullptr;
      if (asm_work_vector_acquire(ctx, &p) != ASM_OK)
        return nullptr;
      v.push_back(p);
      return p;
    }
  };

  // deal.II SolverCG: x_0 = 0, criterion on the unpreconditioned residual
  int
  solve_cg(asm_ctx *ctx, const Preconditioner &pc, const asm_solver_params &sp, double *x, const double *b, asm_solver_result &res,
           std::string &error, double *const *work = nullptr)
  {
    WorkVectors w(ctx);
    double *    r = work ? work[0] : w.get(), *z = work ? work[1] : w.get(), *p = work ? work[2] : w.get(),
           *q = work ? work[3] : w.get();
    if (!r || !z || !p || !q)
      {
        error = "solver: vector allocation failed";
        return ASM_ERR_DEVICE;
      }
    KS_CTX(asm_vec_zero(ctx, x));
    KS_CTX(asm_vec_copy(ctx, r, b));
    double rr = 0., rz = 0., pq = 0.;
    KS_CTX(asm_vec_dot(ctx, r, r, &rr));
    res.initial_residual = res.final_residual = std::sqrt(rr);
    const double tol     = std::max(sp.abs_tolerance, sp.rel_tolerance * res.initial_residual);
    res.n_iterations     = 0;
    res.converged        = res.initial_residual <= tol;
    if (res.converged)
      return ASM_OK;
    KS_TRY(apply_preconditioner(pc, error, z, r));
    KS_CTX(asm_vec_copy(ctx, p, z));
    KS_CTX(asm_vec_dot(ctx, r, z, &rz));
    for (int it = 1; it <= sp.max_iterations; ++it)
      {
        KS_CTX(asm_op_vmult(ctx, q, p));
        KS_CTX(asm_vec_dot(ctx, p, q, &pq));
        const double alpha = rz / pq;
        KS_CTX(asm_vec_axpby(ctx, x, alpha, p, 1.));
        KS_CTX(asm_vec_axpby(ctx, r, -alpha, q, 1.));
        KS_CTX(asm_vec_dot(ctx, r, r, &rr));
        res.n_iterations   = it;
        res.final_residual = std::sqrt(rr);
        if (res.final_residual <= tol)
          {
            res.converged = 1;
            return ASM_OK;
          }
        KS_TRY(apply_preconditioner(pc, error, z, r));
        double rz_new = 0.;
        KS_CTX(asm_vec_dot(ctx, r, z, &rz_new));
        KS_CTX(asm_vec_axpby(ctx, p, 1., z, rz_new / rz));
        rz = rz_new;
      }
    return ASM_OK;
  }

  // deal.II SolverGMRES, classical Gram-Schmidt, restarted with max_n_tmp_vectors - 2 Krylov vectors
  int
  solve_gmres(asm_ctx *ctx, const Preconditioner &pc, const asm_solver_params &sp, double *x, const double *b, asm_solver_result &res,
              std::string &error)
  {
    const int   m     = std::max(1, sp.max_n_tmp_vectors - 2);
    const bool  right = sp.right_preconditioning != 0;
    WorkVectors w(ctx);
    std::vector<double *> V(m + 1);
    for (auto &v : V)
      v = w.get();
    double *t = w.get(), *u = w.get();
    if (!t || !u || !V.back())
      {
        error = "solver: vector allocation failed";
        return ASM_ERR_DEVICE;
      }
    KS_CTX(asm_vec_zero(ctx, x));
    double *r = V[0];
    if (right)
      KS_CTX(asm_vec_copy(ctx, r, b));
    else
      KS_TRY(apply_preconditioner(pc, error, r, b));
    double beta2 = 0.;
    KS_CTX(asm_vec_dot(ctx, r, r, &beta2));
    double beta          = std::sqrt(beta2);
    res.initial_residual = res.final_residual = beta;
    const double tol     = std::max(sp.abs_tolerance, sp.rel_tolerance * beta);
    res.n_iterations     = 0;
    res.converged        = beta <= tol;
    if (res.converged)
      return ASM_OK;
    std::vector<double> H((size_t)(m + 1) * m), cs(m), sn(m), g(m + 1), y(m), hcol(m + 1), hneg(m + 1);
    std::vector<double *> Z;
    bool                  keep_z = right && sp.keep_preconditioned_basis != 0;
    int                 it = 0;
    while (it < sp.max_iterations)
      {
        std::fill(H.begin(), H.end(), 0.);
        std::fill(g.begin(), g.end(), 0.);
        KS_CTX(asm_vec_scale(ctx, V[0], 1. / beta, V[0]));
        g[0]       = beta;
        int k_done = 0;
        for (int k = 0; k < m; ++k)
          {
            // w = A M^-1 v_k (right) or M^-1 A v_k (left), into V[k+1]
            if (right)
              {
                // asm_solver_params::keep_preconditioned_basis: the preconditioned basis vectors are kept (as flexible
                // GMRES does): x += sum_i y_i M^-1 v_i then needs no further application of the preconditioner at the end
                // of the cycle - one V-cycle less per solve.  Default, and without the memory: the vector t is reused and
                // the preconditioner applied to the combination, as deal.II's SolverGMRES does.
                double *zk = t;
                if (keep_z)
                  {
                    if ((int)Z.size() <= k)
                      Z.push_back(w.get());
                    if (Z[k] == nullptr)
                      {
                        keep_z = false;
                        Z.pop_back();
                      }
                    else
                      zk = Z[k];
                  }
                KS_TRY(apply_preconditioner(pc, error, zk, V[k]));
                KS_CTX(asm_op_vmult(ctx, V[k + 1], zk));
              }
            else
              {
                KS_CTX(asm_op_vmult(ctx, t, V[k]));
                KS_TRY(apply_preconditioner(pc, error, V[k + 1], t));
              }
            // classical Gram-Schmidt: all projections from the unmodified w, in one sweep over the basis; the
            // subtraction and the norm of what is left in a second one
            KS_CTX(asm_vec_multi_dot(ctx, V[k + 1], V.data(), k + 1, hcol.data()));
            for (int i = 0; i <= k; ++i)
              {
                H[(size_t)i * m + k] = hcol[i];
                hneg[i]              = -hcol[i];
              }
            double nn = 0.;
            KS_CTX(asm_vec_multi_axpy_norm(ctx, V[k + 1], V.data(), hneg.data(), k + 1, &nn));
            const double hk1 = std::sqrt(nn);
            H[(size_t)(k + 1) * m + k] = hk1;
            if (hk1 > 0.)
              KS_CTX(asm_vec_scale(ctx, V[k + 1], 1. / hk1, V[k + 1]));
            for (int i = 0; i < k; ++i)
              {
                const double a = H[(size_t)i * m + k], c = H[(size_t)(i + 1) * m + k];
                H[(size_t)i * m + k]       = cs[i] * a + sn[i] * c;
                H[(size_t)(i + 1) * m + k] = -sn[i] * a + cs[i] * c;
              }
            const double d = std::hypot(H[(size_t)k * m + k], H[(size_t)(k + 1) * m + k]);
            cs[k]          = H[(size_t)k * m + k] / d;
            sn[k]          = H[(size_t)(k + 1) * m + k] / d;
            H[(size_t)k * m + k]       = d;
            H[(size_t)(k + 1) * m + k] = 0.;
            g[k + 1]                   = -sn[k] * g[k];
            g[k]                       = cs[k] * g[k];
            ++it;
            k_done             = k + 1;
            res.n_iterations   = it;
            res.final_residual = std::fabs(g[k + 1]);
            if (res.final_residual <= tol || it >= sp.max_iterations)
              break;
          }
        // back substitution and update x += [M^-1] sum_i y_i v_i
        for (int i = k_done - 1; i >= 0; --i)
          {
            double s = g[i];
            for (int j = i + 1; j < k_done; ++j)
              s -= H[(size_t)i * m + j] * y[j];
            y[i] = s / H[(size_t)i * m + i];
          }
        if (right && keep_z && (int)Z.size() >= k_done)
          {
            double unused = 0.;
            KS_CTX(asm_vec_multi_axpy_norm(ctx, x, Z.data(), y.data(), k_done, &unused)); // x += sum_i y_i M^-1 v_i
          }
        else
          {
            KS_CTX(asm_vec_zero(ctx, u));
            {
              double unused = 0.;
              KS_CTX(asm_vec_multi_axpy_norm(ctx, u, V.data(), y.data(), k_done, &unused));
            }
            if (right)
              {
                KS_TRY(apply_preconditioner(pc, error, t, u));
                KS_CTX(asm_vec_axpby(ctx, x, 1., t, 1.));
              }
            else
              KS_CTX(asm_vec_axpby(ctx, x, 1., u, 1.));
          }
        if (res.final_residual <= tol)
          {
            res.converged = 1;
            return ASM_OK;
          }
        if (it >= sp.max_iterations)
          return ASM_OK;
        // restart from the true residual
        KS_CTX(asm_op_vmult(ctx, t, x));
        KS_CTX(asm_vec_copy(ctx, u, b));
        KS_CTX(asm_vec_axpby(ctx, u, -1., t, 1.));
        if (right)
          KS_CTX(asm_vec_copy(ctx, V[0], u));
        else
          KS_TRY(apply_preconditioner(pc, error, V[0], u));
        KS_CTX(asm_vec_dot(ctx, V[0], V[0], &beta2));
        beta = std::sqrt(beta2);
      }
    return ASM_OK;
  }

  // ---- V-cycle (deal.II Multigrid::level_v_step) -----------------------------------------------------
  // Dense inverse of the level-0 operator: its columns from the level context's own sweep, constrained rows and
  // columns replaced by the identity, inverted by Cholesky factorisation on the host (the level has a few hundred DoFs).
  constexpr int MAX_COARSE_DIRECT = 2048;

  int
  build_coarse_inverse(asm_mg *mg, std::string &error)
  {
    Level &        lv  = mg->levels[0];
    asm_ctx *const ctx = lv.ctx;
    const int      n   = (int)((long long)lv.layout.owned_size[0] * lv.layout.owned_size[1] * lv.layout.owned_size[2]);
    if ((long long)lv.layout.owned_size[0] * lv.layout.owned_size[1] * lv.layout.owned_size[2] > MAX_COARSE_DIRECT)
      {
        error = "direct coarse-grid solver: the coarsest level has more than " + std::to_string(MAX_COARSE_DIRECT) +
                " DoFs (the numbers of cells must be divisible by a power of two for h-coarsening to get there)";
        return ASM_ERR_NOT_SUPPORTED;
      }
    std::vector<double> A((size_t)n * n), e(n, 0.);
    for (int j = 0; j < n; ++j)
      {
        e[j] = 1.;
        KS_CTX(asm_vector_import(lv.ctx, lv.b, e.data()));
        KS_CTX(asm_op_vmult(lv.ctx, lv.x, lv.b));
        KS_CTX(asm_vector_export(lv.ctx, &A[(size_t)j * n], lv.x));
        e[j] = 0.;
      }
    for (int j = 0; j < n; ++j)
      if (A[(size_t)j * n + j] == 0.) // constrained DoF
        A[(size_t)j * n + j] = 1.;
    // A = L L^T (lower triangle, in place)
    for (int j = 0; j < n; ++j)
      {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k)
          d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.))
          {
            error = "direct coarse-grid solver: the level operator is not positive definite (pure Neumann problem?)";
            return ASM_ERR_NOT_SUPPORTED;
          }
        const double ljj     = std::sqrt(d);
        A[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i)
          {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k)
              v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = v / ljj;
          }
      }
    // columns of the inverse: L y = e_j, L^T x = y (x is row j of the symmetric inverse)
    std::vector<double> inv((size_t)n * n), y(n), U((size_t)n * n); // U = L^T, rows contiguous
    for (int i = 0; i < n; ++i)
      for (int k = 0; k <= i; ++k)
        U[(size_t)k * n + i] = A[(size_t)i * n + k];
    for (int j = 0; j < n; ++j)
      {
        for (int i = 0; i < n; ++i)
          {
            double v = i == j ? 1. : 0.;
            if (i >= j)
              for (int k = j; k < i; ++k)
                v -= A[(size_t)i * n + k] * y[k];
            y[i] = i >= j ? v / A[(size_t)i * n + i] : 0.;
          }
        for (int i = n - 1; i >= 0; --i)
          {
            double v = y[i];
            for (int k = i + 1; k < n; ++k)
              v -= U[(size_t)i * n + k] * inv[(size_t)j * n + k];
            inv[(size_t)j * n + i] = v / A[(size_t)i * n + i];
          }
      }
    if (hipMalloc((void **)&mg->coarse_inverse, inv.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(mg->coarse_inverse, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      {
        error = "direct coarse-grid solver: device allocation failed";
        return ASM_ERR_DEVICE;
      }
    mg->coarse_n = n;
    return ASM_OK;
  }

  // `top`: the level is entered as a level of the multigrid above, i.e. - on the intermediate level - as the coarse-grid
  // problem of the upper multigrid: "n coarse cycles" Richardson steps preconditioned by the lower one
  // (SolverRelaxation under IterationNumberControl, include/multigrid.h:346-365)
  int v_cycle(asm_mg *mg, const int level, double *x, double *b, const bool top);

  // The V-cycle from `level` down (on the hierarchy's own level vectors) as one hipGraph launch.  The capture runs the
  // ordinary code with the level contexts' work directed at a capture stream; nothing is executed by it, the graph is
  // launched on the hierarchy's stream afterwards.  Any failure to capture switches graphs off and runs eagerly.
  int
  coarse_legs_graph(asm_mg *mg, const int level)
  {
    if (mg->graph_state == 0)
      {
        mg->graph_state = 1; // the first V-cycle: eager (kernel attributes, tuned variants are settled by it)
        return v_cycle(mg, level, nullptr, nullptr, true);
      }
    std::vector<double> sig;
    for (int l = 0; l <= level; ++l)
      {
        double st[4];
        asm_internal_cheb_state(mg->levels[l].ctx, st);
        sig.insert(sig.end(), st, st + 4);
      }
    if (mg->graph_exec != nullptr && sig == mg->graph_signature)
      {
        if (hipGraphLaunch(mg->graph_exec, mg->stream) != hipSuccess)
          {
            (void)hipGetLastError();
            mg->graph_state = -1;
            return v_cycle(mg, level, nullptr, nullptr, true);
          }
        ++mg->graph_replays;
        return ASM_OK;
      }
    if (mg->graph_exec != nullptr)
      {
        (void)hipGraphExecDestroy(mg->graph_exec);
        mg->graph_exec = nullptr;
      }
    const auto give_up = [&]() {
      (void)hipGetLastError();
      mg->graph_state = -1;
      return v_cycle(mg, level, nullptr, nullptr, true);
    };
    if (mg->graph_stream == nullptr && hipStreamCreateWithFlags(&mg->graph_stream, hipStreamNonBlocking) != hipSuccess)
      return give_up();
    std::vector<void *> saved(level + 1);
    for (int l = 0; l <= level; ++l)
      saved[l] = asm_internal_swap_stream(mg->levels[l].ctx, (void *)mg->graph_stream);
    const hipStream_t mg_stream = mg->stream;
    mg->stream                  = mg->graph_stream;
    hipGraph_t graph            = nullptr;
    int        err              = ASM_OK;
    const bool began            = hipStreamBeginCapture(mg->graph_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (began)
      {
        err = v_cycle(mg, level, nullptr, nullptr, true);
        if (hipStreamEndCapture(mg->graph_stream, &graph) != hipSuccess)
          graph = nullptr;
      }
    for (int l = 0; l <= level; ++l)
      (void)asm_internal_swap_stream(mg->levels[l].ctx, saved[l]);
    mg->stream = mg_stream;
    if (!began || err != ASM_OK || graph == nullptr)
      {
        if (graph)
          (void)hipGraphDestroy(graph);
        mg->error.clear();
        return give_up();
      }
    const hipError_t inst = hipGraphInstantiate(&mg->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (inst != hipSuccess)
      {
        mg->graph_exec = nullptr;
        return give_up();
      }
    mg->graph_signature = sig;
    if (hipGraphLaunch(mg->graph_exec, mg->stream) != hipSuccess)
      return give_up();
    ++mg->graph_replays;
    return ASM_OK;
  }

  int
  v_cycle(asm_mg *mg, const int level, double *x = nullptr, double *b = nullptr, const bool top = true)
  {
    Level &lv = mg->levels[level];
    // (the finest level works on the caller's vectors)
    x = x ? x : lv.x;
    b = b ? b : lv.b;
    if (top && level == mg->intermediate_level && mg->mp.n_coarse_cycles > 1 && level + 1 < (int)mg->levels.size())
      {
        MG_TRY(v_cycle(mg, level, x, b, false)); // x = M b (from x = 0)
        for (int c = 1; c < mg->mp.n_coarse_cycles; ++c)
          {
            MG_CTX(mg, lv.ctx, asm_op_vmult_fused(lv.ctx, ASM_FUSED_RESIDUAL, mg->rc_r, x, b)); // r = b - A x
            MG_TRY(v_cycle(mg, level, mg->rc_d, mg->rc_r, false));
            MG_CTX(mg, lv.ctx, asm_vec_axpby(lv.ctx, x, 1.0, mg->rc_d, 1.0));                  // x += M r
          }
        return ASM_OK;
      }
    if (level == 0)
      {
        if (mg->mp.coarse_solver == ASM_MG_COARSE_SMOOTHER)
          {
            MG_CTX(mg, lv.ctx, asm_cheb_vmult(lv.ctx, x, b));
            return ASM_OK;
          }
        if (mg->mp.coarse_solver == ASM_MG_COARSE_DIRECT)
          {
            hipLaunchKernelGGL(dense_apply_kernel, dim3(mg->coarse_n), dim3(256), 0, mg->stream, mg->coarse_inverse, mg->coarse_n,
                               box_of(lv.layout), x, b);
            return check_launch(mg, "coarse solve");
          }
        // MGCoarseGridIterativeSolver-like: Jacobi-preconditioned CG (the level-0 context is a diagonal one)
        asm_solver_params sp;
        asm_solver_params_default(&sp);
        sp.type           = ASM_SOLVER_CG;
        sp.rel_tolerance  = mg->mp.coarse_rel_tolerance;
        sp.abs_tolerance  = 0.;
        sp.max_iterations = mg->mp.coarse_max_iterations;
        asm_solver_result res;
        double *const     work[4] = {lv.r, mg->cg_z, mg->cg_p, mg->cg_q};
        const int         err     = solve_cg(lv.ctx, Preconditioner{ASM_SP_INNER, lv.ctx, nullptr}, sp, x, b, res, mg->error, work);
        if (err == ASM_OK && !res.converged)
          ++mg->coarse_not_converged; // (asm_mg_coarse_failures: the V-cycle went on with an inexact coarse correction)
        return err;
      }
    Level &lc = mg->levels[level - 1];
    MG_CTX(mg, lv.ctx, asm_cheb_vmult(lv.ctx, x, b));                                    // pre-smoothing, x_0 = 0
    MG_CTX(mg, lv.ctx, asm_op_vmult_fused(lv.ctx, ASM_FUSED_RESIDUAL, lv.r, x, b));      // r = b - A x
    MG_TRY(asm_mg_restrict(mg, level, lc.b, lv.r));
    if (level - 1 == mg->graph_level && mg->graph_state >= 0)
      MG_TRY(coarse_legs_graph(mg, level - 1));
    else
      MG_TRY(v_cycle(mg, level - 1));
    MG_TRY(asm_mg_prolongate_add(mg, level, x, lc.x));
    if (!mg->mp.one_sided_v_cycle)
      MG_CTX(mg, lv.ctx, asm_cheb_step(lv.ctx, x, b));                                   // post-smoothing
    return ASM_OK;
  }
  int
  estimate_level_eigenvalues(asm_mg *mg)
  {
    for (size_t l = 0; l < mg->levels.size(); ++l)
      {
        if (l == 0 && mg->mp.coarse_solver != ASM_MG_COARSE_SMOOTHER)
          continue;
        double lam = 0.;
        if (const int err = asm_cheb_estimate_eigenvalues(mg->levels[l].ctx, &lam))
          {
            mg->error = std::string("eigenvalue estimate on level ") + std::to_string(l) + ": " + asm_last_error(mg->levels[l].ctx);
            return err;
          }
      }
    return ASM_OK;
  }
} // namespace

extern "C" {

void
asm_mg_params_default(asm_mg_params *p)
{
  std::memset(p, 0, sizeof(*p));
  p->mg_type               = ASM_MG_H;      // "mg type", default "h"
  p->p_sequence            = ASM_MG_BISECT; // "mg p sequence", default "bisect"
  p->one_sided_v_cycle     = 0;
  p->coarse_solver         = ASM_MG_COARSE_AUTO;
  p->coarse_rel_tolerance  = 1e-3;
  p->coarse_max_iterations = 200;
  p->estimate_eigenvalues  = 1;
  p->n_coarse_cycles       = 1; // "n coarse cycles" (include/precondition.h:127)
  p->intermediate_smoother = 0; // "mg intermediate smoother" without a type: the mg smoother (precondition.h:148-158)
  p->intermediate_precond_type = ASM_PRECOND_DIAGONAL;
  p->intermediate_cheb_degree  = 3;
  p->keep_duplicate_level  = 1;
  p->struct_size           = (uint32_t)sizeof(asm_mg_params);
}

void
asm_solver_params_default(asm_solver_params *p)
{
  p->type                  = ASM_SOLVER_GMRES;
  p->max_iterations        = 1000;
  p->abs_tolerance         = 1e-10;
  p->rel_tolerance         = 1e-2;
  p->right_preconditioning = 1;
  p->max_n_tmp_vectors     = 30;
  p->keep_preconditioned_basis = 0;
  p->struct_size = (uint32_t)sizeof(asm_solver_params);
}

const char *
asm_mg_last_error(const asm_mg *mg)
{
  return mg ? mg->error.c_str() : g_mg_create_error.c_str();
}

void
asm_mg_destroy(asm_mg *mg)
{
  if (!mg)
    return;
  const DeviceGuard guard(mg->device);
  (void)hipStreamSynchronize(mg->stream);
  if (!mg->levels.empty() && mg->levels[0].ctx)
    for (double *v : {mg->cg_p, mg->cg_z, mg->cg_q})
      if (v)
        (void)asm_vector_free(mg->levels[0].ctx, v);
  if ((int)mg->levels.size() > mg->intermediate_level && mg->levels[mg->intermediate_level].ctx)
    for (double *v : {mg->rc_r, mg->rc_d})
      if (v)
        (void)asm_vector_free(mg->levels[mg->intermediate_level].ctx, v);
  if (mg->coarse_inverse)
    (void)hipFree(mg->coarse_inverse);
  if (mg->graph_exec)
    (void)hipGraphExecDestroy(mg->graph_exec);
  if (mg->graph_stream)
    (void)hipStreamDestroy(mg->graph_stream);
  for (Level &lv : mg->levels)
    {
      if (lv.ctx)
        {
          for (double *v : {lv.x, lv.b, lv.r})
            if (v)
              (void)asm_vector_free(lv.ctx, v);
          asm_destroy(lv.ctx);
        }
      for (void *p : lv.allocs)
        (void)hipFree(p);
    }
  delete mg;
}

int
asm_mg_create(const asm_params *fine, const asm_mg_params *mp_in, asm_mg **out)
{
  *out = nullptr;
  // (the size is checked through the pointer BEFORE the structure is copied: a full-size copy of a shorter structure
  //  would read past the caller's object.  struct_size is the structure's last member, so the check recognises a
  //  structure that asm_mg_params_default did not fill; it cannot recognise a caller built against a shorter revision.)
  if (mp_in && mp_in->struct_size != (uint32_t)sizeof(asm_mg_params))
    {
      g_mg_create_error = "asm_mg_params: struct_size does not match this library (initialise with asm_mg_params_default)";
      return ASM_ERR_INVALID_VALUE;
    }
  asm_mg_params mp;
  if (mp_in)
    mp = *mp_in;
  else
    asm_mg_params_default(&mp);
  if (fine->world_size < 1 || fine->rank < 0 || fine->rank >= fine->world_size)
    {
      g_mg_create_error = "multigrid: rank / world_size out of range";
      return ASM_ERR_INVALID_VALUE;
    }
  if (mp.coarse_solver < ASM_MG_COARSE_SMOOTHER || mp.coarse_solver > ASM_MG_COARSE_AUTO)
    {
      g_mg_create_error = "unknown coarse-grid solver";
      return ASM_ERR_INVALID_VALUE;
    }
  if (mp.mg_type < ASM_MG_H || mp.mg_type > ASM_MG_PH || mp.p_sequence < ASM_MG_BISECT || mp.p_sequence > ASM_MG_GO_TO_ONE)
    {
      g_mg_create_error = "Multigrid variant / p sequence is not known!";
      return ASM_ERR_INVALID_VALUE;
    }
  if (mp.struct_size != (uint32_t)sizeof(asm_mg_params))
    {
      g_mg_create_error = "asm_mg_params: struct_size does not match this library (initialise with asm_mg_params_default)";
      return ASM_ERR_INVALID_VALUE;
    }
  if (mp.n_coarse_cycles < 1 || (mp.intermediate_smoother != 0 && (mp.intermediate_cheb_degree < 1 ||
                                 (mp.intermediate_precond_type != ASM_PRECOND_FDM && mp.intermediate_precond_type != ASM_PRECOND_DIAGONAL))))
    {
      g_mg_create_error = "n coarse cycles / mg intermediate smoother: invalid value";
      return ASM_ERR_INVALID_VALUE;
    }
  std::unique_ptr<asm_mg> holder(new asm_mg);
  asm_mg *                mg = holder.get();
  mg->mp                     = mp;
  mg->stream                 = (hipStream_t)fine->stream;
  {
    // the device of every level, before anything is created on it (a failure below destroys under this device too)
    int dev = 0;
    (void)hipGetDevice(&dev);
    mg->device = fine->device >= 0 ? fine->device : dev;
  }
  const DeviceGuard guard(mg->device);
  const auto fail            = [&](const int code, const std::string &msg) {
    g_mg_create_error = msg;
    asm_mg_destroy(holder.release());
    return code;
  };
  const std::vector<LevelSpec> specs = level_sequence(*fine, mp);
  const bool                   coarse_auto = mp.coarse_solver == ASM_MG_COARSE_AUTO;
  if (coarse_auto)
    {
      // the dense inverse where level 0 is small enough for it (a linear V-cycle, no host synchronisation inside), else CG
      long long n0 = 1;
      for (int d = 0; d < 3; ++d)
        n0 *= (long long)specs[0].nc[d] * specs[0].p + 1;
      mp.coarse_solver     = n0 <= MAX_COARSE_DIRECT ? ASM_MG_COARSE_DIRECT : ASM_MG_COARSE_CG;
      mg->mp.coarse_solver = mp.coarse_solver;
    }
  mg->levels.resize(specs.size());
  for (size_t l = 0; l < specs.size(); ++l)
    if (specs[l].p == 1)
      mg->intermediate_level = (int)l; // element_centered_preconditioners_01.cc:677-685
  // ---- which levels are block-decomposed (multi-rank): from the finest level down while the blocks stay nested
  const int L = (int)specs.size();
  int       proc_grid[3] = {fine->proc_grid[0], fine->proc_grid[1], fine->proc_grid[2]};
  for (int l = L - 1; l >= 0; --l) // finest level first: it fixes the process grid of the others
    {
      Level &lv = mg->levels[l];
      lv.params = *fine;
      for (int d = 0; d < 3; ++d)
        lv.params.n_cells[d] = specs[l].nc[d];
      lv.params.degree = specs[l].p;
      if (fine->world_size > 1)
        {
          bool dist = l == L - 1;
          if (!dist && l > 0 && mg->levels[l + 1].distributed)
            {
              dist = true;
              for (int d = 0; d < 3; ++d)
                if (specs[l + 1].nc[d] != specs[l].nc[d] && specs[l + 1].nc[d] % (2 * proc_grid[d]) != 0)
                  dist = false; // a rank's cells of level l + 1 are not the children of whole cells of this level
            }
          lv.distributed = dist;
        }
      if (lv.distributed)
        for (int d = 0; d < 3; ++d)
          lv.params.proc_grid[d] = proc_grid[d];
      else
        {
          lv.params.rank       = 0;
          lv.params.world_size = 1;
          lv.params.proc_grid[0] = lv.params.proc_grid[1] = lv.params.proc_grid[2] = 0;
        }
      // "mg intermediate smoother" on the levels of the lower multigrid (include/multigrid.h:268-276)
      if (mp.intermediate_smoother != 0 && l >= 1 && l <= mg->intermediate_level)
        {
          lv.params.precond_type = mp.intermediate_precond_type;
          lv.params.cheb_degree  = mp.intermediate_cheb_degree;
          if (mp.intermediate_precond_type == ASM_PRECOND_DIAGONAL)
            lv.params.element_centric = 1;
        }
      // the coarse CG solve is Jacobi-preconditioned; vertex-star patches need degree >= 2
      if ((l == 0 && mp.coarse_solver != ASM_MG_COARSE_SMOOTHER) || (lv.params.element_centric == 0 && specs[l].p < 2))
        {
          lv.params.precond_type    = ASM_PRECOND_DIAGONAL;
          lv.params.element_centric = 1; // (no patches are built for the diagonal smoother)
        }
      if (l + 1 < L)
        lv.params.kernel_variant = ASM_KERNEL_AUTO; // a kernel family forced for the fine level need not exist on the others
      if (lv.params.geometry == ASM_GEOM_Q2_POINTS)
        return fail(ASM_ERR_NOT_SUPPORTED, "multigrid levels are generated for Cartesian and Kershaw geometry, not for caller-supplied points");
      if (const int err = asm_create(&lv.params, &lv.ctx))
        return fail(err, std::string("level ") + std::to_string(l) + ": " + asm_last_error(nullptr));
      (void)asm_vector_layout(lv.ctx, &lv.layout);
      if (l == L - 1)
        for (int d = 0; d < 3; ++d)
          proc_grid[d] = lv.layout.proc_grid[d];
      if (asm_vector_alloc(lv.ctx, &lv.x) || asm_vector_alloc(lv.ctx, &lv.b) || asm_vector_alloc(lv.ctx, &lv.r))
        return fail(ASM_ERR_DEVICE, "level vector allocation failed");
    }
  {
    asm_ctx *c0 = mg->levels[0].ctx;
    if (asm_vector_alloc(c0, &mg->cg_p) || asm_vector_alloc(c0, &mg->cg_z) || asm_vector_alloc(c0, &mg->cg_q))
      return fail(ASM_ERR_DEVICE, "coarse solver vector allocation failed");
    asm_ctx *ci = mg->levels[mg->intermediate_level].ctx;
    if (mp.n_coarse_cycles > 1 && (asm_vector_alloc(ci, &mg->rc_r) || asm_vector_alloc(ci, &mg->rc_d)))
      return fail(ASM_ERR_DEVICE, "coarse cycle vector allocation failed");
  }
  for (size_t l = 1; l < specs.size(); ++l)
    {
      Level &lv = mg->levels[l];
      const Level &lc = mg->levels[l - 1];
      lv.transfer.fine   = box_of(lv.layout);
      lv.transfer.coarse = box_of(lc.layout);
      for (int d = 0; d < 3; ++d)
        {
          // the rank's window of the level pair (whole levels on one rank)
          Window1D w;
          w.f_begin       = lv.layout.owned_begin[d];
          w.f_size        = lv.layout.owned_size[d];
          w.c_begin       = lc.layout.owned_begin[d];
          const bool top_shared = lc.distributed && lc.layout.proc_coord[d] + 1 < lc.layout.proc_grid[d];
          w.c_size        = lc.layout.owned_size[d] + (top_shared ? 1 : 0);
          w.c_store_begin = lc.layout.owned_begin[d] + lc.layout.store_begin[d];
          w.c_store_end   = w.c_store_begin + lc.layout.store_size[d];
          lv.transfer.coarse.size[d] = w.c_size;
          std::string error;
          if (!build_transfer_1d(lv, lv.params, lc.params, d, w, lv.transfer.d[d], error))
            return fail(ASM_ERR_NOT_SUPPORTED, error);
        }
      // larger levels: one pass per direction (ASM_MG_DIRECT_TRANSFER keeps the single-kernel form everywhere)
      const TransferArgs &t = lv.transfer;
      if ((long long)t.fine.size[0] * t.fine.size[1] * t.fine.size[2] >= (1 << 17) &&
          (long long)t.fine.size[0] * t.fine.size[1] * t.fine.size[2] < (1ll << 31) && std::getenv("ASM_MG_DIRECT_TRANSFER") == nullptr)
        {
          const size_t n_xy = (size_t)t.fine.size[0] * t.fine.size[1] * t.coarse.size[2], n_x = (size_t)t.fine.size[0] * t.coarse.size[1] * t.coarse.size[2];
          void *       q = nullptr;
          if (hipMalloc(&q, (n_xy + n_x) * sizeof(double)) != hipSuccess)
            return fail(ASM_ERR_DEVICE, "transfer work space allocation failed");
          lv.allocs.push_back(q);
          lv.pass_xy = (double *)q;
          lv.pass_x  = lv.pass_xy + n_xy;
        }
    }
  // (ASM_MG_COARSE_AUTO falls back to CG where the dense inverse does not exist: a singular level operator - pure Neumann
  //  problem -, a coarsest level shared by several ranks)
  if (mp.coarse_solver == ASM_MG_COARSE_DIRECT && mg->levels[0].distributed)
    {
      if (!coarse_auto)
        return fail(ASM_ERR_NOT_SUPPORTED, "direct coarse-grid solver: the coarsest level of a multi-rank hierarchy must be a level of its own");
      mp.coarse_solver = mg->mp.coarse_solver = ASM_MG_COARSE_CG;
    }
  if (mp.coarse_solver == ASM_MG_COARSE_DIRECT)
    {
      std::string error;
      if (const int err = build_coarse_inverse(mg, error))
        {
          if (!coarse_auto)
            return fail(err, error);
          mp.coarse_solver = mg->mp.coarse_solver = ASM_MG_COARSE_CG;
        }
    }
  // hipGraph of the small levels' legs (one rank; the coarse CG synchronises with the host and cannot be captured):
  // everything from the highest level of at most 2^20 stored DoFs down, if that is below the finest level.
  // OFF unless ASM_MG_GRAPH is set: measured, the replay is no faster than the plain launches on this stack (V-cycle 1.130
  // against 1.137 ms at 48^3 cells, 2.72 against 2.86 ms at 80^3: the kernels of a graph still follow each other at
  // 5-10 us; profiles/r03_mg_graph.txt)
  if (fine->world_size == 1 && mp.coarse_solver != ASM_MG_COARSE_CG && std::getenv("ASM_MG_GRAPH") != nullptr)
    {
      long long limit = 1ll << 20;
      if (const char *e = std::getenv("ASM_MG_GRAPH_DOFS"))
        limit = std::atoll(e);
      for (int l = L - 2; l >= 1; --l)
        if (mg->levels[l].layout.n_alloc <= limit)
          {
            mg->graph_level = l;
            break;
          }
    }
  // (a multi-rank hierarchy estimates them once it has its communicator: asm_mg_comm_init)
  if (mp.estimate_eigenvalues && fine->world_size == 1)
    if (const int err = estimate_level_eigenvalues(mg))
      return fail(err, mg->error);
  *out = holder.release();
  return ASM_OK;
}

int
asm_mg_graph_replays(const asm_mg *mg, int32_t *level, int64_t *replays)
{
  if (!mg || !level || !replays)
    return ASM_ERR_INVALID_VALUE;
  *level   = mg->graph_state >= 0 ? mg->graph_level : -1;
  *replays = mg->graph_replays;
  return ASM_OK;
}

int
asm_mg_comm_init(asm_mg *mg, const char id[128])
{
  if (!mg || mg->levels.empty())
    return ASM_ERR_INVALID_VALUE;
  const DeviceGuard guard(mg->device);
  Level &           top = mg->levels.back();
  if (!top.distributed)
    {
      mg->error = "asm_mg_comm_init: the hierarchy was created for a single rank (world_size = 1)";
      return ASM_ERR_INVALID_VALUE;
    }
  // ONE communicator (the finest level's) for every block-decomposed level
  MG_CTX(mg, top.ctx, asm_comm_init(top.ctx, id));
  for (Level &lv : mg->levels)
    if (lv.distributed)
      MG_CTX(mg, lv.ctx, asm_internal_comm_share(lv.ctx, top.ctx));
  return mg->mp.estimate_eigenvalues ? estimate_level_eigenvalues(mg) : ASM_OK;
}

// testing aid: the level contexts have been given loopback groups (asm_debug_loopback_init per level, rank r = the
// hierarchy of rank r); called by every rank's host thread: what asm_mg_comm_init does after the communicator exists
int
asm_mg_debug_after_loopback(asm_mg *mg)
{
  if (!mg || mg->levels.empty())
    return ASM_ERR_INVALID_VALUE;
  const DeviceGuard guard(mg->device);
  for (Level &lv : mg->levels)
    if (lv.distributed)
      MG_CTX(mg, lv.ctx, asm_internal_comm_share(lv.ctx, lv.ctx)); // no overlapped exchange; the diagonal smoother's inverse diagonal
  return mg->mp.estimate_eigenvalues ? estimate_level_eigenvalues(mg) : ASM_OK;
}

int
asm_mg_level_distributed(const asm_mg *mg, int32_t level, int32_t *distributed)
{
  if (!mg || !distributed || level < 0 || level >= (int32_t)mg->levels.size())
    return ASM_ERR_INVALID_VALUE;
  *distributed = mg->levels[level].distributed ? 1 : 0;
  return ASM_OK;
}

int
asm_mg_coarse_failures(const asm_mg *mg, int64_t *count)
{
  if (!mg || !count)
    return ASM_ERR_INVALID_VALUE;
  *count = mg->coarse_not_converged;
  return ASM_OK;
}

int
asm_mg_n_levels(const asm_mg *mg, int32_t *n_levels)
{
  *n_levels = (int32_t)mg->levels.size();
  return ASM_OK;
}

int
asm_mg_level_context(asm_mg *mg, int32_t level, asm_ctx **ctx)
{
  if (level < 0 || level >= (int32_t)mg->levels.size())
    {
      mg->error = "asm_mg_level_context: no such level";
      return ASM_ERR_INVALID_VALUE;
    }
  *ctx = mg->levels[level].ctx;
  return ASM_OK;
}

int
asm_mg_prolongate_add(asm_mg *mg, int32_t level, double *dst_fine, const double *src_coarse)
{
  if (level < 1 || level >= (int32_t)mg->levels.size())
    {
      mg->error = "asm_mg_prolongate_add: level must be in [1, n_levels)";
      return ASM_ERR_INVALID_VALUE;
    }
  const DeviceGuard   guard(mg->device);
  const TransferArgs &t = mg->levels[level].transfer;
  const long long     n = (long long)t.fine.size[0] * t.fine.size[1] * t.fine.size[2];
  // a block-decomposed coarse level: the plane shared with the next rank (and whatever else the rows of this rank
  // read) comes from its owner (update_ghost_values); a replicated one is whole on every rank
  if (mg->levels[level - 1].distributed)
    MG_CTX(mg, mg->levels[level - 1].ctx, asm_internal_exchange(mg->levels[level - 1].ctx, const_cast<double *>(src_coarse), 0));
  if (mg->levels[level].pass_xy)
    {
      // x, y, z: coarse -> [fine x][coarse y][coarse z] -> [fine x][fine y][coarse z] -> fine (+=)
      const Level &   lv = mg->levels[level];
      const BoxLayout bx = dense_box(t.fine.size[0], t.coarse.size[1], t.coarse.size[2]),
                      bxy = dense_box(t.fine.size[0], t.fine.size[1], t.coarse.size[2]);
      launch_pass<false>(mg, prolongation_pass(t, 0, t.coarse, bx), lv.pass_x, src_coarse);
      launch_pass<false>(mg, prolongation_pass(t, 1, bx, bxy), lv.pass_xy, lv.pass_x);
      launch_pass<true>(mg, prolongation_pass(t, 2, bxy, t.fine), dst_fine, lv.pass_xy);
      return check_launch(mg, "prolongate_add");
    }
  hipLaunchKernelGGL(prolongate_add_kernel, dim3(launch_blocks(n)), dim3(256), 0, mg->stream, t, dst_fine, src_coarse);
  return check_launch(mg, "prolongate_add");
}

int
asm_mg_restrict(asm_mg *mg, int32_t level, double *dst_coarse, const double *src_fine)
{
  if (level < 1 || level >= (int32_t)mg->levels.size())
    {
      mg->error = "asm_mg_restrict: level must be in [1, n_levels)";
      return ASM_ERR_INVALID_VALUE;
    }
  const DeviceGuard   guard(mg->device);
  const TransferArgs &t = mg->levels[level].transfer;
  const long long     n = (long long)t.coarse.size[0] * t.coarse.size[1] * t.coarse.size[2];
  const Level &       lf = mg->levels[level], &lc = mg->levels[level - 1];
  // Multi-rank: every rank sums over the fine DoFs it owns.  A block-decomposed coarse level: into its owned planes and
  // the plane it shares with the next rank, which goes to its owner and is added there (compress(add); every other ghost
  // entry must be zero for that).  A replicated coarse level below a block-decomposed fine one: into the whole level,
  // summed over the ranks afterwards.
  const bool junction = lf.distributed && !lc.distributed;
  if (lc.distributed)
    MG_CTX(mg, lc.ctx, asm_internal_zero_ghosts(lc.ctx, dst_coarse));
  if (lf.pass_xy)
    {
      // z, y, x: fine -> [fine x][fine y][coarse z] -> [fine x][coarse y][coarse z] -> coarse
      const BoxLayout bx = dense_box(t.fine.size[0], t.coarse.size[1], t.coarse.size[2]),
                      bxy = dense_box(t.fine.size[0], t.fine.size[1], t.coarse.size[2]);
      launch_pass<false>(mg, restriction_pass(t, 2, t.fine, bxy), lf.pass_xy, src_fine);
      launch_pass<false>(mg, restriction_pass(t, 1, bxy, bx), lf.pass_x, lf.pass_xy);
      launch_pass<false>(mg, restriction_pass(t, 0, bx, t.coarse), dst_coarse, lf.pass_x);
    }
  else
    hipLaunchKernelGGL(restrict_kernel, dim3(launch_blocks(n)), dim3(256), 0, mg->stream, t, dst_coarse, src_fine);
  MG_TRY(check_launch(mg, "restrict"));
  if (lc.distributed)
    MG_CTX(mg, lc.ctx, asm_internal_exchange(lc.ctx, dst_coarse, 1));
  else if (junction)
    {
      // (the padding of the rows is part of the allocation: whatever it holds is summed too and never read)
      MG_CTX(mg, lf.ctx, asm_internal_allreduce(lf.ctx, dst_coarse, lc.layout.n_alloc));
    }
  return ASM_OK;
}

int
asm_mg_vmult(asm_mg *mg, double *dst, const double *src)
{
  const DeviceGuard guard(mg->device);
  Level &           top = mg->levels.back();
  if (dst == src)
    {
      MG_CTX(mg, top.ctx, asm_vec_copy(top.ctx, top.b, src));
      return v_cycle(mg, (int)mg->levels.size() - 1, dst, top.b);
    }
  return v_cycle(mg, (int)mg->levels.size() - 1, dst, const_cast<double *>(src)); // (one rank: src is only read)
}

int
asm_solve(asm_ctx *ctx, asm_mg *mg, int preconditioner, const asm_solver_params *sp_in, double *x, const double *b,
          asm_solver_result *result)
{
  asm_solver_params sp;
  if (sp_in)
    sp = *sp_in;
  else
    asm_solver_params_default(&sp);
  std::string  local_error;
  std::string &error = mg ? mg->error : local_error;
  if (sp_in && sp_in->struct_size != (uint32_t)sizeof(asm_solver_params))
    {
      error = "asm_solver_params of another ABI revision: fill it with asm_solver_params_default first";
      return ASM_ERR_INVALID_VALUE;
    }
  if (preconditioner == ASM_SP_MULTIGRID)
    {
      if (mg == nullptr)
        return ASM_ERR_INVALID_VALUE;
      asm_ctx *fine = mg->levels.back().ctx;
      if (ctx != nullptr && ctx != fine)
        {
          error = "asm_solve: with ASM_SP_MULTIGRID the context must be the finest level of the multigrid object";
          return ASM_ERR_INVALID_VALUE;
        }
      ctx = fine;
    }
  if (ctx == nullptr || result == nullptr || x == nullptr || b == nullptr || x == b)
    return ASM_ERR_INVALID_VALUE;
  std::memset(result, 0, sizeof(*result));
  const Preconditioner pc{preconditioner, ctx, mg};
  // the right-hand side with zeros on constrained DoFs (AffineConstraints::set_zero, as op.rhs() leaves it)
  WorkVectors w(ctx);
  double *    rhs = w.get();
  if (rhs == nullptr || asm_vec_copy(ctx, rhs, b) || asm_vec_constraints_set_zero(ctx, rhs))
    {
      error = asm_last_error(ctx);
      return ASM_ERR_DEVICE;
    }
  if (sp.type == ASM_SOLVER_CG)
    return solve_cg(ctx, pc, sp, x, rhs, *result, error);
  if (sp.type == ASM_SOLVER_GMRES)
    return solve_gmres(ctx, pc, sp, x, rhs, *result, error);
  error = "Solver <" + std::to_string(sp.type) + "> is not known!";
  return ASM_ERR_INVALID_VALUE;
}
} // extern "C"
