// Launch interfaces of the HIP kernels (all asynchronous on the given stream).
#pragma once

#include "grid.h"

#include <hip/hip_runtime.h>

#include <array>
#include <vector>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#  error "the kernels are written for gfx950 (wave64): their shuffle offsets, per-wave partial sums and lane maps assume 64 lanes"
#endif

namespace asmamd
{
  // ---- generic sweeps (kernels_generic.hip): dst must be zeroed by the caller,
  //      contributions are added with fp64 atomics
  // tmp == nullptr: contributions are added to dst with atomics (dst zeroed by the caller);
  // otherwise tmp holds n_cells * m^3 doubles: patch results are stored there and summed per DoF
  // by a second kernel that writes every stored entry of dst (no zeroing, no atomics)
  // fu (with tmp only): the fused vector update applied by the summing kernel, TiledMode PC_* (declared below)
  struct FdmFused
  {
    int           mode = 2; // PC_PLAIN
    const double *x = nullptr, *x_old = nullptr;
    double        f1 = 0., f2 = 1.;
    // without tmp: 1 = every DoF has exactly one patch with a non-zero weight (restricted additive Schwarz on
    // element-centred patches, one rank): the patch kernel stores the updated value itself - no zeroing, no atomics,
    // no update passes; constrained DoFs are set by vec_constrained_fused afterwards
    // 2 = beside fdm_slice7_acc: the listed cells outside the box add f2 * contribution with atomics, constrained DoFs skipped
    int direct = 0;
  };
  bool fdm_generic(const DevGrid &g, const DevFdm &f, const double *src, double *dst, double *tmp, const FdmFused &fu, hipStream_t st);
  // kernels_slice7.hip: the patch solves of the box f.s7_* into tmp[cell][343] (false: not applicable)
  bool fdm_slice7(const DevGrid &g, const DevFdm &f, const double *src, double *tmp, hipStream_t st);
  // the whole 7^3 sweep without tmp (f.s7_flag set, one rank): dst = the fused update fu of P^-1 src; the box's patch
  // results are summed in LDS while a workgroup marches along z, the cells outside the box and the DoF planes between
  // two workgroups' columns are added with atomics (false: not applicable)
  bool fdm_slice7_acc(const DevGrid &g, const DevFdm &f, const double *src, double *dst, const FdmFused &fu, hipStream_t st);
  // dst += sign * A src (sign = -1 on a dst initialised with b gives the residual b - A src in one pass)
  bool laplace_generic(const DevGrid &g, const DevOp &op, const double *src, double *dst, bool diag, hipStream_t st, double sign = 1.0);

  // ---- low-order sweeps, one thread per cell (kernels_small.hip); same contracts as fdm_generic's solve stage
  //      (store: dst = tmp[cell][m^3]; otherwise atomics on dst) and laplace_generic; false: not applicable
  bool fdm_cell(const DevGrid &g, const DevFdm &f, const double *src, double *dst, bool store, hipStream_t st);
  bool laplace_cell(const DevGrid &g, const DevOp &op, const double *src, double *dst, bool diag, hipStream_t st, double sign);

  // ---- tiled owner-computes sweeps (kernels_tiled.hip), degree 4, n_overlap 1
  enum TiledMode
  {
    // Laplace operator
    OP_PLAIN    = 0, // dst = A src                         (constrained rows: 0)
    OP_RESIDUAL = 1, // dst = b - A src                     (constrained rows: identity)
    // FDM preconditioner
    PC_PLAIN  = 2, // dst = P^-1 src                        (constrained rows: 0)
    PC_FIRST  = 3, // dst = f2 * P^-1 src                   (constrained rows: identity)
    PC_STEP1  = 4, // dst = x + f2 * P^-1 src
    PC_UPDATE = 5  // dst = x + f1 (x - x_old) + f2 * P^-1 src ; dst may alias x_old
  };
  struct TiledArgs
  {
    int           mode;
    const double *src;
    double *      dst;
    const double *b;     // OP_RESIDUAL
    const double *x;     // PC_STEP1 / PC_UPDATE
    const double *x_old; // PC_UPDATE
    double        f1, f2;
    // overlap of the ghost exchange with the sweep: 0 = all workgroups, 1 = only those that produce
    // DoFs a neighbouring rank ghosts (cells 0 and nc-1 of a direction whose face is an interface),
    // 2 = the others
    int phase       = 0;
    int iface[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    // host side only: the device the launch goes to and its number of compute units (no process-global caches)
    int device = 0;
    int n_cu   = 0;
    // Laplace sweep: cell rows (waves) per workgroup of the second-generation kernel (9, 10, 11), 0 = first generation
    int rows = 0;
    // single-phase launches: work index of every launch position (tiled_order), or nullptr = the positions themselves
    const int *order = nullptr;
    // the same for the wave-pair sweep (pair_order), and its number of z chunks (0: the schedule model's choice)
    const int *pair_order = nullptr;
    // ... and of its two-phase launches: the first pair_n_first entries are phase 1 (pair_phase_order)
    const int *pair_phase_order = nullptr;
    int        pair_n_first     = 0;
    int        pair_n_march = 0, pair_order_offset = 0; // set by the launcher (kernels_pair.hip: launch_pair)
    int        pair_zchunks = 0;
    int        pair_waves   = 0; // waves per tile of the wave-pair sweep: 2, or 0 / 4 = four where the mode has them
  };
  // Work order of a single-phase tiled sweep (device-independent, host side).  The workgroups at Dirichlet faces live
  // up to 1.5 times as long as the others (constrained rows, full-matrix sweeps; profiles/r03_workgroup_trace.txt), and a
  // launch of three rounds of workgroups ended 11-12 % after the average compute unit had finished.  order[p] = work
  // index (tile x fastest, then tile y, then z chunk) taken by launch position p, where every XCD keeps its contiguous range of
  // work indices (L2 locality) but takes the expensive ones first.  Returns the number of positions (0: nothing to reorder).
  int tiled_order(const DevGrid &g, int n_cu, int rows, std::vector<int> &order);
  // the tiling itself (host side): out = {tiles x, tiles y, z chunks, layers per chunk, first cell of tile 0 in x, in y,
  // owned cell columns per tile, owned cell rows per tile}
  void tiled_plan(const DevGrid &g, int n_cu, int rows, int out[8]);
  bool laplace_tiled(const DevGrid &g, const DevOp &op, const TiledArgs &a, hipStream_t st);
  bool fdm_tiled(const DevGrid &g, const DevFdm &f, const TiledArgs &a, hipStream_t st);

  // ---- wave-pair sweep (kernels_pair.hip), degree 4, cell patches / box cells on tensor-product meshes: two waves share a
  //      tile of 8 x 8 cells (one lane per cell), march along z and apply the Kronecker-diagonal cell operator of the
  //      tables k with the fused update of a.mode; single-phase launches only (false: not applicable, nothing launched).
  //      The march neither reads nor writes constrained DoFs; further workgroups of the same launch set them (identity rows).
  bool pair_sweep(const DevGrid &g, const DevKron &k, const TiledArgs &a, hipStream_t st);
  // work order of the pair sweep (host side, device-independent): order[p] = work index (tile x fastest, tile y, z chunk)
  // of launch position p: every XCD keeps its contiguous range of the work but starts with its expensive workgroups (tiles
  // at physical boundaries).  Returns the number of workgroups.
  int pair_order(const DevGrid &g, int n_cu, std::vector<int> &order, int zchunks = 0);
  // chunk counts in the order of the schedule model's estimate (best first): the context measures the first few
  void pair_chunk_candidates(const DevGrid &g, int n_cu, std::vector<int> &zchunks);
  // the tiling of the pair sweep (host side): out = {tiles x, tiles y, z chunks, layers per chunk, first cell x, first cell y}
  void pair_plan(const DevGrid &g, int n_cu, int out[6], int zchunks = 0);
  void pair_tiles(const DevGrid &g, int n_cu, std::vector<std::array<int, 6>> &tiles, int zchunks = 0);
  int  pair_phase_order(const DevGrid &g, int n_cu, const bool iface[3][2], std::vector<int> &order, int zchunks = 0);

  // ---- vector kernels (kernels_vector.hip)
  // elementwise over the whole allocation
  void vec_residual(double *r, const double *b, long long n, hipStream_t st);                // r = b - r
  void vec_scale(double *y, const double *x, double a, long long n, hipStream_t st);         // y = a x
  void vec_mul(double *y, const double *x, const double *d, long long n, hipStream_t st);    // y = x .* d
  void vec_axpby(double *y, double a, const double *x, double b, long long n, hipStream_t st); // y = a x + b y
  // y = x + f1 (x - xo) + f2 t ; y may alias xo
  void vec_cheb_update(double *y, const double *x, const double *xo, const double *t, double f1, double f2, long long n,
                       hipStream_t st);
  void vec_invert_diagonal(double *d, long long n, hipStream_t st); // |d|>1e-10 ? 1/d : 1
  // reductions over the owned box; result[0] (device) accumulated, must be zeroed before
  void vec_dot(const DevGrid &g, const double *x, const double *y, double *result, hipStream_t st);
  // up to 8 vectors of one Gram-Schmidt step: result[i] += <w, v[i]> (owned box); w += sum_i a[i] v[i] (whole allocation)
  // with result[0] += |w_new|^2 over the owned box if result is given
  struct VecList
  {
    const double *v[8];
    double        a[8];
    int           n;
  };
  void vec_multi_dot(const DevGrid &g, const double *w, const VecList &L, double *result, hipStream_t st);
  void vec_multi_axpy(const DevGrid &g, double *w, const VecList &L, long long n_alloc, double *result, hipStream_t st);
  // deal.II set_initial_guess: v = (global lexicographic index % 11)
  void vec_initial_guess(const DevGrid &g, double *v, const long long gbegin[3], const long long gsize[3], hipStream_t st);
  // v(owned) += a
  void vec_add_scalar_owned(const DevGrid &g, double *v, double a, hipStream_t st);
  // Dirichlet planes: mode 0: dst = 0 ; 1: dst = src ; 2: dst = b - x
  void vec_constrained(const DevGrid &g, int mode, double *dst, const double *src, const double *b, hipStream_t st);
  // dst = b, on Dirichlet planes b - x (levels of fewer than 2^22 stored DoFs: one launch where a copy and vec_constrained
  // would be two; every stored entry, the padding of the rows is left alone)
  void vec_residual_init(const DevGrid &g, double *dst, const double *b, const double *x, hipStream_t st);
  // Dirichlet planes: dst = the fused update of FdmFused applied to the unit row, i.e. with P^-1 src replaced by src (0 for PC_PLAIN)
  void vec_constrained_fused(const DevGrid &g, const FdmFused &fu, double *dst, const double *src, hipStream_t st);
  // measurement aid: dst = src for n 16-byte vectors, one 16 KB tile per workgroup, four non-temporal 16-byte loads in
  // flight per lane before the first store: the copy rate these boxes sustain (tools/microbench_copy.hip: 6.2 TB/s,
  // the guide's 6.29), i.e. the achievable HBM rate next to the 8 TB/s peak
  void vec_stream_copy(void *dst, const void *src, long long n_vec16, hipStream_t st);
  // box copy between a packed buffer and a vector (halo exchange, import/export)
  // mode 0: buf <- vec ; 1: vec <- buf ; 2: vec += buf ; 3: vec <- 0 (buf unused)
  // false (nothing launched): unknown mode, missing operand or a box outside the stored extent
  bool vec_box(const DevGrid &g, int mode, double *vec, double *buf, const int lo[3], const int size[3], hipStream_t st);
  // the same for up to 26 boxes in one launch: box b is copied to / from buf + offset[b]
  struct BoxList
  {
    int       n;
    int       lo[26][3], size[26][3];
    long long offset[26];
  };
  bool vec_boxes(const DevGrid &g, int mode, double *vec, double *buf, const BoxList &boxes, hipStream_t st);
} // namespace asmamd
