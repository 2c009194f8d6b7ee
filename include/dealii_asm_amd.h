/*
 * dealii_asm_amd.h -- C ABI of the MI355X-native matrix-free Laplace operator and
 * FDM additive-Schwarz / Chebyshev smoother (libdealii_asm_amd.so).
 *
 * Every entry point names the reference interface (peterrum/dealii-asm, paths
 * relative to the reference root) it replaces.  The reference has no FFI of its
 * own: its components are glued by C++ templates, so the
 * "binding" a maintainer adds is the thin C++ shim in dealii_asm_amd.hpp, shown
 * in INTEGRATION.md.
 *
 * Conventions
 *  - all vectors are DEVICE pointers to double, laid out as described by
 *    asm_vector_layout() (a lexicographic box, x fastest, with optional ghost
 *    planes and a padded row pitch); the caller owns them;
 *  - all calls are asynchronous on the stream given at creation unless stated;
 *    one in-flight call per context (the reference's operators hold mutable
 *    scratch and are not re-entrant either: include/matrix_free.h:1545-1564);
 *  - return value 0 = success, otherwise an ASM_ERR_* code; asm_last_error()
 *    returns the message (the reference throws C++ exceptions instead:
 *    include/precondition.templates.h:26,79-81,815).
 */
#ifndef DEALII_ASM_AMD_H
#define DEALII_ASM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct asm_ctx asm_ctx;

enum asm_status
{
  ASM_OK                = 0,
  ASM_ERR_INVALID_VALUE = 1,
  ASM_ERR_NOT_SUPPORTED = 2, /* mirrors ExcNotImplemented() in the reference */
  ASM_ERR_DEVICE        = 3, /* HIP / RCCL runtime error */
  ASM_ERR_NO_DEVICE     = 4  /* no HIP device: the product path never falls back to the CPU */
};

/* boundary type of one face of the global box (include/matrix_free.h:350-360
 * passes boundary id {1} as Dirichlet, {2} as Neumann; periodic faces come from
 * matrix_free_loop_08.likwid.cc:163-169) */
enum asm_boundary
{
  ASM_BC_PERIODIC  = 0,
  ASM_BC_DIRICHLET = 1,
  ASM_BC_NEUMANN   = 2
};

/* Restrictors::WeightingType, include/restrictors.h:8-15 (same order) */
enum asm_weighting
{
  ASM_W_NONE = 0,
  ASM_W_PRE  = 1,
  ASM_W_POST = 2,
  ASM_W_RAS  = 3,
  ASM_W_SYMM = 4
};

/* mesh generators of the reference drivers */
enum asm_geometry
{
  ASM_GEOM_CARTESIAN = 0, /* subdivided box, matrix_free_loop_08.likwid.cc:160-174 */
  ASM_GEOM_KERSHAW   = 1, /* include/kershaw.h:39-80 via MappingQCache(2),
                             element_centered_preconditioners_01.cc:352-424 */
  ASM_GEOM_Q2_POINTS = 2  /* caller supplies the 27 mapping support points per cell */
};

/* "type" of the inner preconditioner of the Chebyshev smoother,
 * include/precondition.templates.h:481-573 */
enum asm_preconditioner
{
  ASM_PRECOND_FDM      = 0,
  ASM_PRECOND_DIAGONAL = 1
};

/* "ev algorithm", include/precondition.templates.h:113-133 */
enum asm_ev_algorithm
{
  ASM_EV_AUTO            = 0, /* lanczos if symmetric else power iteration */
  ASM_EV_LANCZOS         = 1,
  ASM_EV_POWER_ITERATION = 2
};

/* "polynomial type" of PreconditionChebyshev, include/precondition.templates.h:135-152 */
enum asm_polynomial_type
{
  ASM_POLY_FIRST_KIND  = 0, /* "1st kind" (default) */
  ASM_POLY_FOURTH_KIND = 1  /* "4th kind" */
};

/* The fused vector updates PreconditionChebyshev (and relaxation / multigrid drivers) hand to
 * vmult(dst, src, pre, post) as host lambdas over index ranges (include/operator.h:1367-1430,
 * include/matrix_free.h:960-986, MFRunner::loop include/matrix_free_internal.h:309-359).  Host code
 * cannot run inside a GPU sweep, so the closed set of updates those lambdas carry is applied in the
 * kernels' store phase instead: this enum is the one place where the reference's surface is narrowed. */
enum asm_fused_update
{
  ASM_FUSED_NONE       = 0, /* dst = Op src                                  (constrained rows: 0)        */
  ASM_FUSED_RESIDUAL   = 1, /* operator only:       dst = b - A src          (constrained rows: identity) */
  ASM_FUSED_CHEB_FIRST = 2, /* preconditioner only: dst = f2 P^-1 src        (x_0 = 0)                    */
  ASM_FUSED_RELAX      = 3, /* preconditioner only: dst = f1 x + f2 P^-1 src (damped relaxation, first
                               Chebyshev step from x with f1 = 1)                                        */
  ASM_FUSED_CHEB_NEXT  = 4  /* preconditioner only: dst = x + f1 (x - x_old) + f2 P^-1 src; dst may alias x_old */
};

/* kernel family used for the sweeps */
enum asm_kernel_variant
{
  ASM_KERNEL_AUTO    = 0,
  ASM_KERNEL_GENERIC = 1, /* one patch per thread group, scatter-add with fp64 atomics */
  ASM_KERNEL_TILED   = 2  /* owner-computes tiles, fused Chebyshev updates, no atomics */
};

typedef struct asm_params
{
  /* --- discretisation ------------------------------------------------- */
  int32_t degree;         /* FE_Q(p), 1 <= p <= 7 ("degree") */
  int32_t n_cells[3];     /* GLOBAL number of cells per direction */
  int32_t boundary[3][2]; /* asm_boundary of the low/high face per direction */
  int32_t geometry;       /* asm_geometry */
  double  extent[3];      /* edge lengths of the (undeformed) box */
  double  kershaw_eps[2]; /* eps_y, eps_z ("mesh.eps") */
  const double *q2_points; /* ASM_GEOM_Q2_POINTS: host array [cell][27][3], cells x-fastest */

  /* --- FDM preconditioner: keys of create_fdm_preconditioner,
   *     include/precondition.templates.h:195-211 ------------------------ */
  int32_t precond_type;     /* asm_preconditioner */
  int32_t n_overlap;        /* "n overlap", default 1, clamped to p */
  int32_t weighting_type;   /* "weighting type" */
  int32_t element_centric;  /* "element centric": 1 = cell patches, 0 = vertex-star */

  /* --- Chebyshev: create_chebyshev_preconditioner,
   *     include/precondition.templates.h:108-152 ------------------------ */
  int32_t cheb_degree;      /* "degree", default 3 */
  double  smoothing_range;  /* "smoothing range", default 20 */
  int32_t eig_n_iterations; /* eig_cg_n_iterations, 40 in the reference */
  int32_t ev_algorithm;     /* asm_ev_algorithm */

  /* --- execution ------------------------------------------------------- */
  int32_t kernel_variant; /* asm_kernel_variant */
  int32_t device;         /* HIP device ordinal, -1 = current */
  void   *stream;         /* hipStream_t, NULL = default stream */

  /* --- domain decomposition (one process per GPU) ---------------------- */
  int32_t rank;
  int32_t world_size;
  int32_t proc_grid[3]; /* ranks per direction, product == world_size; 0 = choose */

  /* --- appended in ABI revision 2 --------------------------------------- */
  int32_t polynomial_type; /* asm_polynomial_type, "polynomial type", default "1st kind" */

  /* --- appended in ABI revision 3 --------------------------------------- */
  uint32_t struct_size; /* set by asm_params_default; every entry point that takes an asm_params refuses any other value (a
                         * caller built against another revision of this header is told so instead of being read past its
                         * end) */
} asm_params;

/* description of the device vector layout (replaces
 * LinearAlgebra::distributed::Vector + Utilities::MPI::Partitioner as used in
 * include/operator.h:1491-1498 and include/matrix_free.h:994-998) */
typedef struct asm_layout
{
  int64_t n_global_dofs; /* LaplaceOperatorMatrixFree::m(), include/operator.h:767-771 */
  int64_t n_owned;       /* locally owned DoFs */
  int64_t n_ghost;       /* stored entries owned by another rank (or periodic images) */
  int64_t n_alloc;       /* doubles to allocate per vector (includes row padding) */
  int32_t owned_begin[3]; /* first owned DoF plane, GLOBAL logical coordinate */
  int32_t owned_size[3];  /* owned DoF planes per direction */
  int32_t store_begin[3]; /* first stored plane, LOCAL logical coordinate (<= 0) */
  int32_t store_size[3];  /* stored planes per direction */
  int64_t stride[3];      /* element strides of the box */
  int32_t cell_begin[3];  /* first local cell, global cell coordinate */
  int32_t cell_size[3];   /* local cells per direction */
  int32_t proc_grid[3];
  int32_t proc_coord[3];
} asm_layout;

/* one direction of the ghost exchange: boxes in LOCAL logical DoF coordinates
 * (subtract asm_layout::store_begin for storage indices); index [0] is the low
 * neighbour, [1] the high neighbour; peer < 0 or size 0 = nothing to do.
 * Both sides post their sends in the order low, high and their receives in the
 * order high, low. */
typedef struct asm_exchange_step
{
  int32_t direction;
  int32_t peer[2];
  int32_t send_lo[2][3], send_size[2][3];
  int32_t recv_lo[2][3], recv_size[2][3];
} asm_exchange_step;

/* one message of the single-round ghost update (update_ghost_values of the tiled kernels):
 * the rank exchanges one box with each of its up to 26 face, edge and corner neighbours;
 * boxes in LOCAL logical DoF coordinates, offsets (in doubles) into one send and one
 * receive buffer.  Sends are posted in the order of the array, receives in reverse order. */
typedef struct asm_direct_message
{
  int32_t peer;      /* rank of the neighbour */
  int32_t offset[3]; /* its block coordinate minus mine, each in {-1, 0, 1} */
  int32_t send_lo[3], send_size[3];
  int32_t recv_lo[3], recv_size[3];
  int64_t send_offset, recv_offset, send_count, recv_count;
} asm_direct_message;

/* ----- life cycle ------------------------------------------------------- */

/* fills *p with the reference's defaults */
void asm_params_default(asm_params *p);
/* Sets one option by the reference's JSON key and string value (host only, no GPU):
 *   "n overlap", "weighting type" = none|pre|post|symm|ras, "element centric" = true|false
 *                                              (create_fdm_preconditioner, precondition.templates.h:195-211)
 *   "type" = FDM|Diagonal, "degree" (of the Chebyshev polynomial), "smoothing range",
 *   "ev algorithm" = lanczos|power iteration, "polynomial type" = 1st kind|4th kind
 *                                              (create_chebyshev_preconditioner, :108-152, 481-573)
 *   "fe degree" (the top-level "degree" of the experiment files: polynomial degree of FE_Q)
 *   "weight sequence" = global|compressed|dg, "reuse partitioner", "overlap pre post" = true|false, "optimize" = 0..3:
 *     implementation choices of the reference that leave the numbers unchanged (:195-211, 341, 388, 534): validated and
 *     accepted, no effect here; "sub mesh approximation": only the default 3 (other values: ASM_ERR_NOT_SUPPORTED)
 * Unknown keys or values return ASM_ERR_INVALID_VALUE, where the reference throws
 * (precondition.templates.h:26, 79-81, 130-133, 148-152). */
int asm_params_set(asm_params *p, const char *key, const char *value);

/* ASPoissonPreconditioner::ASPoissonPreconditioner (include/matrix_free.h:73-894),
 * LaplaceOperatorMatrixFree::setup_mapping_and_indices (include/operator.h:490-753)
 * and create_chebyshev_preconditioner without the eigenvalue estimate. */
int asm_create(const asm_params *p, asm_ctx **out);
void asm_destroy(asm_ctx *ctx);
const char *asm_last_error(const asm_ctx *ctx); /* ctx may be NULL: last creation error */

/* host-only description of what rank `rank` would own; needs no GPU
 * (used by the multi-process CPU tests) */
int asm_describe_layout(const asm_params *p, asm_layout *out);

/* host-only: the (up to 3) steps of update_ghost_values (add = 0) or of
 * compress(add) (add = 1, received values are ADDED, sent ghost planes zeroed
 * afterwards) that rank `p->rank` executes; this is the plan asm_update_ghost_values
 * / asm_compress_add run over RCCL.  Needs no GPU. */
int asm_exchange_plan(const asm_params *p, int add, asm_exchange_step steps[3], int32_t *n_steps);
/* the single-round form of update_ghost_values (no GPU needed): up to 26 messages */
int asm_exchange_plan_direct(const asm_params *p, asm_direct_message msgs[26], int32_t *n_msgs);

/* RCCL bootstrap: rank 0 fills a 128-byte id, the host broadcasts it, every rank
 * joins.  Replaces the MPI communicator held by deal.II's Partitioner. */
int asm_comm_unique_id(char id[128]);
int asm_comm_init(asm_ctx *ctx, const char id[128]);

/* ----- vectors ---------------------------------------------------------- */

int asm_vector_layout(const asm_ctx *ctx, asm_layout *out);
/* initialize_dof_vector (include/operator.h:1491-1498): allocates and zeroes */
int asm_vector_alloc(asm_ctx *ctx, double **v);
int asm_vector_free(asm_ctx *ctx, double *v);
/* copy between a HOST array holding this rank's owned DoFs in lexicographic
 * order (x fastest, owned_size box) and a device vector (synchronous) */
int asm_vector_import(asm_ctx *ctx, double *dev, const double *host_owned);
int asm_vector_export(asm_ctx *ctx, double *host_owned, const double *dev);
/* VectorDataExchange::update_ghost_values_start/finish
 * (include/matrix_free_internal.h:13-47) */
int asm_update_ghost_values(asm_ctx *ctx, double *v);
/* compress_start/finish with VectorOperation::add + zero_out_ghost_values
 * (include/matrix_free_internal.h:49-103) */
int asm_compress_add(asm_ctx *ctx, double *v);

/* LinearAlgebra::distributed::Vector operations the reference's Krylov solvers and multigrid call on
 * its vectors (deal.II SolverCG / SolverGMRES / Multigrid: operator*, add, sadd, equ, operator=):
 * all on the context's stream; elementwise ones cover the whole allocation (ghost planes included,
 * which keeps ghost values consistent), the inner product runs over the owned DoFs and is summed
 * over all ranks (synchronous: the value is returned to the host). */
int asm_vec_dot(asm_ctx *ctx, const double *x, const double *y, double *result); /* x * y */
/* One Gram-Schmidt step of SolverGMRES in two sweeps over the vectors (deal.II's solver_gmres.h does the same for
 * LinearAlgebra::distributed::Vector: all inner products of a step in one pass, the subtraction and the norm of the
 * result in a second one): result[i] = w * v[i];  w += sum_i a[i] v[i], *norm2 = w * w afterwards. */
int asm_vec_multi_dot(asm_ctx *ctx, const double *w, const double *const *v, int32_t n, double *result);
int asm_vec_multi_axpy_norm(asm_ctx *ctx, double *w, const double *const *v, const double *a, int32_t n, double *norm2);
int asm_vec_axpby(asm_ctx *ctx, double *y, double a, const double *x, double b); /* y = a x + b y (sadd) */
int asm_vec_scale(asm_ctx *ctx, double *y, double a, const double *x);           /* y = a x (equ); y may be x */
int asm_vec_copy(asm_ctx *ctx, double *y, const double *x);                      /* y = x */
int asm_vec_zero(asm_ctx *ctx, double *y);                                       /* y = 0 */
int asm_vec_constraints_set_zero(asm_ctx *ctx, double *y); /* AffineConstraints::set_zero: y = 0 on Dirichlet DoFs */

/* ----- operators -------------------------------------------------------- */

/* LaplaceOperatorMatrixFree::vmult(dst, src), include/operator.h:1353-1365 */
int asm_op_vmult(asm_ctx *ctx, double *dst, const double *src);
/* LaplaceOperatorMatrixFree::compute_inverse_diagonal, include/operator.h:1512-1524 */
int asm_op_inverse_diagonal(asm_ctx *ctx, double *out);
/* ASPoissonPreconditioner::vmult(dst, src), include/matrix_free.h:925-954
 * (or DiagonalMatrix::vmult when precond_type == ASM_PRECOND_DIAGONAL) */
int asm_precond_vmult(asm_ctx *ctx, double *dst, const double *src);

/* The overloads with pre/post operations, LaplaceOperatorMatrixFree::vmult(dst, src, pre, post)
 * (include/operator.h:1367-1430) and ASPoissonPreconditioner::vmult(dst, src, pre, post)
 * (include/matrix_free.h:960-986), for the updates of asm_fused_update.  An external Chebyshev,
 * relaxation or multigrid driver reaches the fused, atomics-free sweeps through these two; what
 * asm_cheb_vmult / asm_cheb_step do internally is exactly a sequence of such calls.
 *   b:     ASM_FUSED_RESIDUAL
 *   x:     ASM_FUSED_RELAX, ASM_FUSED_CHEB_NEXT;  x_old: ASM_FUSED_CHEB_NEXT (dst may alias it)
 * dst must not alias src. */
int asm_op_vmult_fused(asm_ctx *ctx, int mode, double *dst, const double *src, const double *b);
int asm_precond_vmult_fused(asm_ctx *ctx, int mode, double *dst, const double *src, const double *x, const double *x_old,
                            double f1, double f2);

/* PreconditionChebyshev::estimate_eigenvalues as driven by
 * include/precondition.templates.h:458-463; synchronous.  Returns the
 * largest-eigenvalue estimate (before the 1.2 safety factor). */
int asm_cheb_estimate_eigenvalues(asm_ctx *ctx, double *lambda_max);
/* install an eigenvalue estimate obtained elsewhere */
int asm_cheb_set_eigenvalue(asm_ctx *ctx, double lambda_max);
/* PreconditionerBase::vmult / step as reached through PreconditionerAdapter
 * (include/preconditioners.h:725-742, 897-926) and WrapperForGMG
 * (include/precondition.h:24-78): dst = p_k(P^-1 A) P^-1 src with a zero
 * (vmult) or the given (step) initial guess in dst. */
int asm_cheb_vmult(asm_ctx *ctx, double *dst, const double *src);
int asm_cheb_step(asm_ctx *ctx, double *dst, const double *src);

/* ----- queries ---------------------------------------------------------- */

/* ASPoissonPreconditioner::n_fdm_instances, include/matrix_free.h:1000-1004
 * (number of distinct 1-D eigen-decompositions stored) */
int asm_n_fdm_instances(const asm_ctx *ctx, int64_t *n);
/* ASPoissonPreconditioner::memory_consumption, include/matrix_free.h:988-992
 * (bytes of per-cell / per-instance device data of operator + preconditioner) */
int asm_memory_consumption(const asm_ctx *ctx, int64_t *bytes);
/* theta, delta, lambda_max of the Chebyshev polynomial in use */
int asm_cheb_parameters(const asm_ctx *ctx, double out[3]);
/* ASPoissonPreconditioner::is_symmetric (include/matrix_free.h:896-904): 1 for the weightings none and symm
 * (and for the diagonal preconditioner), 0 otherwise */
int asm_is_symmetric(const asm_ctx *ctx, int32_t *symmetric);
/* kernel variant actually selected (asm_kernel_variant) */
int asm_kernel_variant_in_use(const asm_ctx *ctx, int32_t *variant);
/* hipStreamSynchronize on the context's stream */
int asm_synchronize(asm_ctx *ctx);
/* per-kernel timing for the roofline report: while enabled, every launch of the
 * Laplace sweep kernel (kernel 0) and of the FDM sweep kernel (kernel 1) is
 * bracketed by HIP events on the context's stream (the LIKWID marker regions of
 * matrix_free_loop_08.likwid.cc:365-377 are the reference's counterpart) */
int asm_profile_enable(asm_ctx *ctx, int on);
int asm_profile_get(asm_ctx *ctx, int kernel, int64_t *launches, double *total_ms);
/* Measurement aid ("peak_measured" of bench.py): rate of a plain device-to-device copy between two arrays of `bytes` bytes
 * (allocated and freed by the call; read + written bytes per second, best of `reps` launches): what the device sustains
 * next to the 8 TB/s HBM peak the roofline fractions are quoted against */
int asm_measure_copy_bandwidth(asm_ctx *ctx, int64_t bytes, int32_t reps, double *gb_per_s);
/* Host-only (no GPU): the tiling of the degree-4 owner-computes sweeps for the block `p` describes, on a device of n_cu
 * compute units, for the sweep variant of `rows` cell rows per workgroup (0 or 8: first generation = the FDM sweep; 9-11: the
 * second-generation Laplace sweeps).  plan = {tiles x, tiles y, z chunks, layers per chunk, first cell of tile 0 in x, in y
 * (-1: a cell lies below the block's first one, 0: physical boundary), owned cell columns, owned cell rows per tile};
 * order[i] = work index taken by launch position i of a single-phase launch (*n_order entries, 0: launch order = work
 * order).  Diagnostic: lets the tiling be checked without a device (tests/test_abi_cpu.py). */
int asm_tiled_plan(const asm_params *p, int32_t n_cu, int32_t rows, int32_t plan[8], int32_t *order, int32_t max_order, int32_t *n_order);
/* The same for the wave-pair sweeps (degree 4 on Cartesian cells; csrc/kernels_pair.hip): plan = {tiles x, tiles y, z chunks,
 * layers per chunk, first cell of tile 0 in x, in y} of the main tiles; a tile = 8 x 8 cells of which 7 x 7 (all 8 where tile 0
 * starts at cell 0) own their DoFs, the cell behind the block's last one owns the block's last DoF plane. */
int asm_pair_plan(const asm_params *p, int32_t n_cu, int32_t plan[6], int32_t *order, int32_t max_order, int32_t *n_order);
/* ... and its tiles (plan[0] x plan[1] main tiles first, then the strip tiles that replace a last tile column / row of three
 * cells or less): tiles[i] = {first cell x, y, cells in x, in y (8 x 8, 4 x 16 or 16 x 4), bit 0 / 1: the tile owns its first
 * column / row (else they are its halo), last cell column it may own}; the work index of asm_pair_plan's order is
 * tile + n_tiles * z chunk. */
int asm_pair_tiles(const asm_params *p, int32_t n_cu, int32_t (*tiles)[6], int32_t max_tiles, int32_t *n_tiles);
/* which sweeps of a tiled context run on the wave-pair kernels (measured against the tiled kernels when the context is
 * created; ASM_PAIR=1 / 0 fixes it): *laplace, *fdm = 1 / 0.  Replaces the same reference methods as the tiled kernels:
 * LaplaceOperatorMatrixFree::vmult (include/operator.h:1367-1430), ASPoissonPreconditioner::vmult (include/matrix_free.h:1023-1062). */
int asm_pair_sweeps_in_use(const asm_ctx *ctx, int32_t *laplace, int32_t *fdm);
/* z chunks of the context's wave-pair sweeps: the schedule model (asm_pair_plan) ranks the chunk counts, the context
 * measures the first few when it is created; 0 = the model's first choice, ASM_PAIR_ZCHUNKS=n fixes it.  Reported by bench.py. */
int asm_pair_zchunks(const asm_ctx *ctx, int32_t *zchunks);
/* waves per tile of the context's wave-pair sweeps: 4 (two waves per SIMD, 128 + 128 registers each) or 2 (one wave per
 * SIMD); measured with the chunk counts, ASM_PAIR_WAVES=2|4 fixes it.  The three-term update always runs with 2. */
int asm_pair_waves(const asm_ctx *ctx, int32_t *waves);
/* Cell rows (waves) per workgroup of the context's tiled Laplace sweep: 8 = first generation, 9 / 10 / 11 = second
 * (picked by a measurement at creation, or ASM_TILED_ROWS); 0 = no tiled Laplace sweep.  Reported by bench.py. */
int asm_tiled_laplace_rows(const asm_ctx *ctx, int32_t *rows);
/* timing helper for benchmarks: runs `reps` back-to-back calls of one entry
 * point between two HIP events on the context's stream and returns the
 * elapsed milliseconds.  what: 0 op_vmult, 1 precond_vmult, 2 cheb_vmult, 3 cheb_step */
/* Promise of the caller: between two calls vectors are modified only through this library
 * (asm_vector_import, vmult, step).  A vector produced by the last sweep of a Chebyshev call then
 * keeps the ghost values its overlapped exchange has filled, and the next call does not exchange
 * it again.  Off by default: vmult/step then update the ghosts of their input themselves, as
 * LinearAlgebra::distributed::Vector::update_ghost_values does inside MFRunner::loop
 * (include/matrix_free_internal.h:321-324). */
int asm_set_keep_ghosts(asm_ctx *ctx, int on);

int asm_time_calls(asm_ctx *ctx, int what, int reps, double *dst, const double *src, float *ms);

/* ----- the smoother's callers: multigrid V-cycle, transfers, outer Krylov solvers ----------------------
 * (single rank: level hierarchies are not block-decomposed yet, world_size > 1 is refused) */

typedef struct asm_mg asm_mg;

/* "mg type", element_centered_preconditioners_01.cc:552-553, 640-672 */
enum asm_mg_type
{
  ASM_MG_H  = 0, /* geometric coarsening at the fine degree (default) */
  ASM_MG_P  = 1, /* polynomial coarsening on the fine mesh */
  ASM_MG_HP = 2, /* p levels on the coarsest mesh below the h levels */
  ASM_MG_PH = 3  /* h levels at the lowest degree below the p levels (the experiments' choice) */
};
/* "mg p sequence", element_centered_preconditioners_01.cc:555-586 */
enum asm_mg_p_sequence
{
  ASM_MG_BISECT          = 0,
  ASM_MG_DECREASE_BY_ONE = 1,
  ASM_MG_GO_TO_ONE       = 2
};
/* "mg coarse grid solver": the reference hands any system preconditioner to MGCoarseGridApplyPreconditioner
 * (include/multigrid.h:264-265, 352-358; Trilinos ML AMG in the experiments, precondition.templates.h:592-613) */
enum asm_mg_coarse
{
  ASM_MG_COARSE_SMOOTHER = 0, /* the level-0 Chebyshev smoother's vmult (tests/.../small fixtures) */
  ASM_MG_COARSE_CG       = 1, /* Jacobi-preconditioned CG to a relative tolerance (in place of AMG) */
  ASM_MG_COARSE_DIRECT   = 2, /* dense inverse of the level operator, built at creation (levels of <= 2048 DoFs): one
                               * kernel per coarse solve and a linear V-cycle, also in place of AMG */
  ASM_MG_COARSE_AUTO     = 3  /* default: DIRECT where the coarsest level has at most 2048 DoFs and its operator a dense inverse
                               * (not: pure Neumann problem, level shared by several ranks), CG otherwise (a CG coarse solve
                               * makes the V-cycle a non-linear preconditioner and synchronises with the host every iteration) */
};

typedef struct asm_mg_params
{
  int32_t mg_type;           /* asm_mg_type */
  int32_t p_sequence;        /* asm_mg_p_sequence */
  int32_t one_sided_v_cycle; /* "one-sided v-cycle": identity post-smoother (include/multigrid.h:305-322) */
  int32_t coarse_solver;     /* asm_mg_coarse */
  double  coarse_rel_tolerance;   /* ASM_MG_COARSE_CG */
  int32_t coarse_max_iterations;
  int32_t estimate_eigenvalues; /* 1: run asm_cheb_estimate_eigenvalues on every level at creation */
  /* ---- ABI revision 3 (round 3): the two nested hierarchies of include/multigrid.h:260-375 ------------------
   * The 'ph' / 'hp' level lists name the joint level twice (element_centered_preconditioners_01.cc:656-671); both copies
   * are levels here too (identity transfer between them, a smoother each).  The levels up to the last one of degree 1
   * ("intermediate level", :677-685) are smoothed by "mg intermediate smoother" and form the coarse-grid solver of the
   * multigrid over the levels above, applied "n coarse cycles" times as a Richardson iteration (multigrid.h:346-365). */
  int32_t n_coarse_cycles;            /* "n coarse cycles", default 1 (precondition.h:127) */
  int32_t intermediate_smoother;      /* 0: "mg intermediate smoother" has no type = the mg smoother (precondition.h:148-158);
                                         1: Chebyshev of the fields below */
  int32_t intermediate_precond_type;  /* asm_precond_type of its inner preconditioner, e.g. ASM_PRECOND_DIAGONAL */
  int32_t intermediate_cheb_degree;   /* its "degree" */
  int32_t keep_duplicate_level;       /* default 1; 0: the joint level once (the hierarchy of ABI revisions 1-2) */
  uint32_t struct_size;               /* set by asm_mg_params_default; asm_mg_create rejects any other value */
} asm_mg_params;

void asm_mg_params_default(asm_mg_params *p);
/* Level hierarchy of `fine` (cells, degree, smoother configuration; rank 0 of 1): level operators and
 * smoothers are contexts of this library ("mg smoother" = the smoother `fine` describes, on every level with
 * n_overlap clamped to the level's degree; vertex-star patches become the diagonal smoother at degree 1),
 * MGTwoLevelTransfer between neighbouring levels (include/multigrid.h:289-299, 336-346). */
int  asm_mg_create(const asm_params *fine, const asm_mg_params *mp, asm_mg **out);
void asm_mg_destroy(asm_mg *mg);
const char *asm_mg_last_error(const asm_mg *mg);
/* Multi-rank hierarchies (fine->world_size > 1; every rank calls asm_mg_create with its rank): a level is block-decomposed
 * by the fine level's process grid as long as every rank's cells of the finer level are the children of its cells there
 * (always for p-coarsening; h-coarsening: cells per direction divisible by twice the ranks of that direction); the levels
 * below - at least level 0 - are held whole by every rank.  Transfers exchange the coarse vector's shared planes
 * (prolongation) or add the parts of the shared coarse planes at their owner (restriction; an all-reduce into the first
 * replicated level), include/multigrid.h:289-346.  asm_mg_comm_init creates ONE RCCL communicator for all levels (collective,
 * same unique id on every rank: asm_comm_unique_id) and then estimates the levels' eigenvalues (estimate_eigenvalues).
 * Unmeasured between GPUs so far: verified with all ranks as contexts of one process (tests/test_gpu_partitioned_mg.py). */
int asm_mg_comm_init(asm_mg *mg, const char id[128]);
int asm_mg_level_distributed(const asm_mg *mg, int32_t level, int32_t *distributed);
int asm_mg_n_levels(const asm_mg *mg, int32_t *n_levels);
/* level 0 = coarsest; the context of level n_levels-1 is the fine operator the outer solver works on */
int asm_mg_level_context(asm_mg *mg, int32_t level, asm_ctx **ctx);
/* MGTwoLevelTransfer between level (fine) and level-1 (coarse):
 *   prolongate_and_add: dst_fine += P src_coarse;  restrict_and_add into a zeroed vector: dst_coarse = P^T src_fine
 * (constrained DoFs of either level neither receive nor contribute) */
int asm_mg_prolongate_add(asm_mg *mg, int32_t level, double *dst_fine, const double *src_coarse);
int asm_mg_restrict(asm_mg *mg, int32_t level, double *dst_coarse, const double *src_fine);
/* PreconditionMG::vmult (include/multigrid.h:466-472): one V-cycle from a zero initial guess; dst, src: vectors of
 * the finest level.  Pre-smoothing = smoother vmult, post-smoothing = smoother step (deal.II MGSmootherRelaxation) */
int asm_mg_vmult(asm_mg *mg, double *dst, const double *src);
/* Number of coarse-grid CG solves (ASM_MG_COARSE_CG) that ended at coarse_max_iterations without reaching
 * coarse_rel_tolerance since the hierarchy was created: the V-cycles went on with an inexact coarse correction (the
 * reference's AMG V-cycle, include/precondition.templates.h:592-613, is inexact by construction and reports nothing) */
int asm_mg_coarse_failures(const asm_mg *mg, int64_t *count);
/* hipGraph of the V-cycle's small levels (one rank, coarse solver other than ASM_MG_COARSE_CG; only if the environment
 * variable ASM_MG_GRAPH is set when the hierarchy is created - measured no faster than the plain launches on ROCm 7.2, so
 * off by default): the legs from *level down - the highest level of at most 2^20 stored DoFs below the finest - are captured
 * on the second V-cycle and replayed as one graph launch while the levels' Chebyshev parameters stand; *level = -1: no graph
 * (not requested, not applicable, or the capture was refused and the hierarchy went back to plain launches); *replays = graph
 * launches so far.  Not in the reference (CPU); results are those of the plain launches (to the summation order of the low-order sweeps' atomics). */
int asm_mg_graph_replays(const asm_mg *mg, int32_t *level, int64_t *replays);

/* solve(), element_centered_preconditioners_01.cc:108-263: SolverCG / SolverGMRES under ReductionControl */
enum asm_solver_type
{
  ASM_SOLVER_CG    = 0,
  ASM_SOLVER_GMRES = 1
};
/* create_system_preconditioner, include/precondition.templates.h:249-620 (the types this path covers) */
enum asm_system_preconditioner
{
  ASM_SP_IDENTITY  = 0,
  ASM_SP_INNER     = 1, /* the context's FDM / Diagonal preconditioner */
  ASM_SP_CHEBYSHEV = 2, /* the context's Chebyshev smoother (vmult) */
  ASM_SP_MULTIGRID = 3  /* one V-cycle of the asm_mg given to asm_solve */
};
typedef struct asm_solver_params
{
  int32_t type;                  /* "type": CG | GMRES */
  int32_t max_iterations;        /* "max iterations", 1000 */
  double  abs_tolerance;         /* "abs tolerance", 1e-10 */
  double  rel_tolerance;         /* "rel tolerance", 1e-2 */
  int32_t right_preconditioning; /* "use right preconditioning", true */
  int32_t max_n_tmp_vectors;     /* "max n tmp vectors", 30 (Krylov vectors per restart = this - 2) */
  /* GMRES with right preconditioning: 0 (default) = deal.II's SolverGMRES, which applies the preconditioner once more to the
   * combination of the basis at the end of a cycle; 1 = keep the preconditioned basis vectors M^-1 v_i (as SolverFGMRES does):
   * one preconditioner application less per cycle for as many more work vectors; the same iterates for a linear preconditioner */
  int32_t keep_preconditioned_basis;
  uint32_t struct_size; /* set by asm_solver_params_default; asm_solve refuses any other value */
} asm_solver_params;
typedef struct asm_solver_result
{
  int32_t n_iterations; /* SolverControl::last_step() */
  int32_t converged;
  double  initial_residual, final_residual;
} asm_solver_result;
void asm_solver_params_default(asm_solver_params *p);
/* x = 0 initially (as solve() sets it); ctx: the fine-level context (for ASM_SP_MULTIGRID: level n_levels-1 of mg,
 * ctx may then be NULL); classical Gram-Schmidt in GMRES, the residual estimate of the Givens recurrence
 * decides convergence, as in deal.II */
int asm_solve(asm_ctx *ctx, asm_mg *mg, int preconditioner, const asm_solver_params *sp, double *x, const double *b,
              asm_solver_result *result);

/* ----- testing aids ---------------------------------------------------------
 * All ranks of a partition as contexts of ONE process on ONE device: RCCL cannot
 * connect two ranks on the same GPU, so the tests use these two calls to run the
 * partitioned kernels, vector layouts and exchange plans on real device buffers.
 * asm_debug_external_exchange(ctx, 1): vmult/step no longer exchange ghosts
 * themselves.  asm_debug_exchange_all: the exchange MFRunner::loop performs
 * (include/matrix_free_internal.h:309-359; add == 0: update_ghost_values,
 * add != 0: compress(add)) for vecs[r] of ctxs[r], r = rank = 0..n-1, at once. */
int asm_debug_external_exchange(asm_ctx *ctx, int on);
/* tiled sweeps as the overlapped exchange launches them (boundary workgroups, then interior
 * workgroups), without the exchange */
int asm_debug_two_phase(asm_ctx *ctx, int on);
/* ctxs[r] = rank r of n in this process; afterwards every rank is driven by its own host thread and
 * vmult/step/estimate run their multi-rank control flow (streams, events, launch phases, ghost
 * bookkeeping) with device copies + host barriers in place of RCCL */
int asm_debug_loopback_init(asm_ctx **ctxs, int n);
/* ... of a multi-rank multigrid hierarchy: after asm_debug_loopback_init on the contexts of every block-decomposed level
 * (ctxs[r] = that level of rank r's hierarchy), every rank's host thread calls this in place of asm_mg_comm_init */
int asm_mg_debug_after_loopback(asm_mg *mg);
int asm_debug_exchange_all(asm_ctx **ctxs, double **vecs, int n, int add);

#ifdef __cplusplus
}
#endif
#endif /* DEALII_ASM_AMD_H */
