// Context, C ABI, Chebyshev driver and ghost exchange.
//
// Host-side mirror of the reference's operator stack for the hot path:
//   LaplaceOperatorMatrixFree         include/operator.h:266-1628
//   ASPoissonPreconditioner           include/matrix_free.h:63-1568
//   MFRunner / VectorDataExchange     include/matrix_free_internal.h:3-364
//   PreconditionChebyshev (deal.II)   configured at include/precondition.templates.h:89-158
#include "../../include/dealii_asm_amd.h"
#include "kernels.h"
#include "kron.h"
#include "setup.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include <map>

#ifdef ASM_STAMP
namespace asmamd
{
  void tiled_debug_stamps(unsigned long long out[32], bool reset);
}
#endif

using namespace asmamd;

static thread_local std::string g_create_error;

// Testing aid: the ranks of a partition as contexts of ONE process, driven by one host thread per
// rank (asm_debug_loopback_init).  Send/recv become device-to-device copies between the ranks'
// buffers, fenced by host barriers; streams, events, launch phases and ghost bookkeeping are
// the ones of the RCCL path.
struct LoopbackGroup
{
  std::vector<asm_ctx *>  ranks;
  std::vector<double>     scalars;
  std::vector<const double *> buffers; // all-reduce of device buffers (multigrid: replicated coarse levels)
  std::mutex              m;
  std::condition_variable cv;
  int                     waiting = 0, generation = 0;
  void
  barrier()
  {
    std::unique_lock<std::mutex> lk(m);
    const int                    gen = generation;
    if (++waiting == (int)ranks.size())
      {
        waiting = 0;
        ++generation;
        cv.notify_all();
      }
    else
      cv.wait(lk, [&] { return gen != generation; });
  }
};

struct asm_ctx
{
  HostSetup                s;
  std::unique_ptr<Basis1D> basis;
  hipStream_t              stream = nullptr;
  int                      device = 0; // HIP device ordinal every call of this context runs on
  int                      n_cu   = 0; // its compute units (tile scheduling of the tiled sweeps)
  int                      laplace_rows = 0; // TiledArgs::rows of the tiled Laplace sweep (tune_tiled_laplace)
  const int *              order_a = nullptr, *order_p = nullptr; // work order of the single-phase tiled sweeps (tiled_order): A, P
  DevGrid                  g;
  DevFdm                   f;
  DevOp                    op;
  // wave-pair sweeps (kernels_pair.hip) of a tiled context on a tensor-product mesh: tables of A and of P (valid = 0: the
  // tiled kernels take the sweep); ASM_PAIR=0 switches them off
  DevKron                  kron_a{}, kron_p{};
  const int *              order_pair = nullptr; // work order of the wave-pair sweeps (pair_order)
  const int *              order_pair_phase = nullptr; // ... of their two-phase launches (pair_phase_order)
  int                      pair_n_first     = 0;
  bool                     pair_a = false, pair_p = false; // which sweeps take the wave-pair kernel (tune_pair)
  int                      pair_zchunks = 0;               // ... with how many z chunks (0: the schedule model's choice)
  int                      pair_waves = 0;                 // ... and waves per tile (2, or 0 / 4: four where the update has them)
  std::vector<void *>      allocs;
  bool                     ras_direct = true; // ASM_RAS_ATOMICS: restricted additive Schwarz through the atomics path on one rank too
  std::vector<double *>    work_pool; // released solver work vectors (asm_work_vector_*), kept until asm_destroy
  long long                data_bytes = 0;
  double *                 inv_diag   = nullptr;
  double *                 R = nullptr, *T = nullptr, *S = nullptr; // scratch vectors
  double *                 d_scalar = nullptr;
  double *                 h_scalar = nullptr; // pinned
  double                   lambda_max = 0., theta = 0., delta = 0.;
  bool                     ev_set     = false;
  ncclComm_t               comm       = nullptr;
  bool                     comm_shared = false; // the communicator belongs to another context (multigrid levels share the finest level's)
  std::shared_ptr<LoopbackGroup> loop; // testing aid, in place of comm
  double *                 hbuf[4]    = {nullptr, nullptr, nullptr, nullptr}; // send lo/hi, recv lo/hi
  long long                hbuf_size  = 0;
  bool                     external_exchange = false; // testing aid: ghosts are exchanged by asm_debug_exchange_all
  bool                     debug_two_phase   = false; // testing aid: boundary and interior workgroups in two launches
  // single-round ghost update (update_ghost_values of the tiled kernels): setup.h: build_direct_plan
  asm_direct_message       dmsg[26];
  int                      n_dmsg = 0;
  BoxList                  dsend, drecv;
  double *                 dbuf[2] = {nullptr, nullptr}; // send, receive
  double *                 patch_tmp = nullptr; // generic FDM: n_cells * m^3 patch results (kernels.h: fdm_generic)
  // Overlap of the ghost exchange with the sweep that produces the vector (tiled kernels, inside
  // PreconditionChebyshev::vmult/step): the workgroups that produce DoFs a neighbour ghosts run first, the
  // exchange of the NEW vector then runs on comm_stream beside the remaining workgroups, and the next
  // sweep only waits for ev_exchanged.
  hipStream_t              comm_stream  = nullptr;
  hipStream_t              bnd_stream   = nullptr; // boundary workgroups of an overlapped sweep (high priority)
  hipEvent_t               ev_boundary  = nullptr, ev_exchanged = nullptr, ev_start = nullptr;
  bool                     overlap      = false;   // available (set by asm_comm_init)
  bool                     in_cheb      = false;   // inside cheb_run
  bool                     last_sweep   = false;   // the sweep's result is not read by another sweep of this call
  bool                     keep_ghosts  = false;   // asm_set_keep_ghosts: vectors are not modified between calls
  const double *           exchanged_vec = nullptr; // vector whose ghosts an overlapped exchange has filled / is filling
  bool                     exchange_pending = false; // ... and ctx->stream has not waited for it yet
  // optional per-kernel timing with HIP events on the context's stream
  bool                                           profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[2];
  size_t                                         prof_used[2] = {0, 0};
  std::string              error;
};

#define HIP_TRY_CTX(c_, expr)                                                                      \
  do                                                                                               \
    {                                                                                              \
      const hipError_t e_ = (expr);                                                                \
      if (e_ != hipSuccess)                                                                        \
        {                                                                                          \
          (c_)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
          return ASM_ERR_DEVICE;                                                                   \
        }                                                                                          \
    }                                                                                              \
  while (0)
// a box copy the launcher refused (kernels.h: vec_box / vec_boxes): unknown mode, missing buffer, box outside the extent
#define BOX_TRY_CTX(c_, expr)                                                                      \
  do                                                                                               \
    {                                                                                              \
      if (!(expr))                                                                                 \
        {                                                                                          \
          (c_)->error = "box copy refused (mode / buffer / extent): " #expr;                       \
          return ASM_ERR_INVALID_VALUE;                                                            \
        }                                                                                          \
    }                                                                                              \
  while (0)
#define BOX_TRY(expr) BOX_TRY_CTX(ctx, expr)
#define HIP_TRY(expr)                                                                              \
  do                                                                                               \
    {                                                                                              \
      const hipError_t e_ = (expr);                                                                \
      if (e_ != hipSuccess)                                                                        \
        {                                                                                          \
          ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
          return ASM_ERR_DEVICE;                                                                   \
        }                                                                                          \
    }                                                                                              \
  while (0)

#define NCCL_TRY(expr)                                                                             \
  do                                                                                               \
    {                                                                                              \
      const ncclResult_t e_ = (expr);                                                              \
      if (e_ != ncclSuccess)                                                                       \
        {                                                                                          \
          ctx->error = std::string(#expr) + ": " + ncclGetErrorString(e_);                         \
          return ASM_ERR_DEVICE;                                                                   \
        }                                                                                          \
    }                                                                                              \
  while (0)

namespace
{
  // Every entry point runs on the context's device, whatever the calling thread's current device is,
  // and leaves the thread's current device as it found it.
  struct DeviceScope
  {
    int  prev = -1;
    bool switched = false;
    explicit DeviceScope(const asm_ctx *ctx)
    {
      if (ctx != nullptr && hipGetDevice(&prev) == hipSuccess && prev != ctx->device)
        switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~DeviceScope()
    {
      if (switched)
        (void)hipSetDevice(prev);
    }
  };

  // A rejected launch (LDS limit, bad grid, other architecture) must not leave the call looking
  // successful: fetch the runtime's sticky launch error after the launches of an operation.
  int
  launch_status(asm_ctx *ctx, const char *where)
  {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess)
      return ASM_OK;
    ctx->error = std::string("kernel launch failed in ") + where + ": " + hipGetErrorString(e);
    return ASM_ERR_DEVICE;
  }

  template <typename T>
  int
  upload(asm_ctx *ctx, const std::vector<T> &h, const T **d)
  {
    *d = nullptr;
    if (h.empty())
      return ASM_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, h.size() * sizeof(T)));
    ctx->allocs.push_back(p);
    ctx->data_bytes += h.size() * sizeof(T);
    HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = (const T *)p;
    return ASM_OK;
  }

  // gives a table of upload() back (tables of candidates a measurement did not pick); nothing may be using it any more
  template <typename T>
  void
  release(asm_ctx *ctx, const T *&d)
  {
    if (d == nullptr)
      return;
    const auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), (void *)d);
    if (it != ctx->allocs.end())
      {
        (void)hipFree(*it);
        ctx->allocs.erase(it);
      }
    d = nullptr;
  }

  // the work-order tables of the wave-pair sweeps for a number of z chunks (0: the schedule model's): single-phase order
  // and the order of the two-phase launches; the tables they replace must not be in use any more
  int
  upload_pair_orders(asm_ctx *ctx, const int zchunks)
  {
    release(ctx, ctx->order_pair);
    release(ctx, ctx->order_pair_phase);
    std::vector<int> order;
    if (std::getenv("ASM_NO_TILED_ORDER") == nullptr)
      {
        pair_order(ctx->g, ctx->n_cu, order, zchunks);
        if (const int err = upload(ctx, order, &ctx->order_pair))
          return err;
      }
    bool iface[3][2];
    for (int d = 0; d < 3; ++d)
      for (int f = 0; f < 2; ++f)
        iface[d][f] = ctx->s.face[d][f] == FACE_INTERFACE;
    ctx->pair_n_first = pair_phase_order(ctx->g, ctx->n_cu, iface, order, zchunks);
    return upload(ctx, order, &ctx->order_pair_phase);
  }

  int
  alloc_vec(asm_ctx *ctx, double **v)
  {
    HIP_TRY(hipMalloc((void **)v, ctx->s.n_alloc * sizeof(double)));
    HIP_TRY(hipMemsetAsync(*v, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    return ASM_OK;
  }

  bool
  multi_rank(const asm_ctx *ctx)
  {
    return ctx->s.prm.world_size > 1 || ctx->s.self_interface;
  }

  // kernel ids: 0 = Laplace sweep kernel, 1 = FDM sweep kernel
  void
  prof_mark(asm_ctx *ctx, const int id, const bool begin)
  {
    if (!ctx->profile)
      return;
    auto &ev = ctx->prof_events[id];
    if (begin)
      {
        if (ctx->prof_used[id] == ev.size())
          {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
              return;
            ev.emplace_back(a, b);
          }
        (void)hipEventRecord(ev[ctx->prof_used[id]].first, ctx->stream);
      }
    else if (ctx->prof_used[id] < ev.size())
      {
        (void)hipEventRecord(ev[ctx->prof_used[id]].second, ctx->stream);
        ++ctx->prof_used[id];
      }
  }

  // ---- ghost exchange ----------------------------------------------------
  // dimension-by-dimension exchange of slabs (x, then y with the x ghosts, then z),
  // the structured-block equivalent of Partitioner::export_to_ghosted_array
  // (include/matrix_free_internal.h:13-47); add = true runs the reverse
  // (import_from_ghosted_array with VectorOperation::add, :49-84) and zeroes
  // the ghost planes afterwards (:86-103).
  int exchange_loopback(asm_ctx *ctx, double *v, hipStream_t st, bool add = false);

  int
  exchange(asm_ctx *ctx, double *v, const bool add)
  {
    if (!multi_rank(ctx) || ctx->external_exchange)
      return ASM_OK;
    if (ctx->comm == nullptr && !ctx->loop)
      {
        ctx->error = "world_size > 1 but asm_comm_init() has not been called";
        return ASM_ERR_INVALID_VALUE;
      }
    // whatever runs on comm_stream has to be finished before this stream reads the ghosts it fills
    // or issues another RCCL operation
    if (ctx->exchange_pending)
      {
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_exchanged, 0));
        ctx->exchange_pending = false;
      }
    if (!add && ctx->exchanged_vec == v)
      return ASM_OK; // filled by the overlapped exchange that followed the sweep which produced v
    if (add && ctx->exchanged_vec == v)
      ctx->exchanged_vec = nullptr; // compress(add) changes owned values and zeroes the ghost planes
    const DevGrid &g = ctx->g;
    if (ctx->loop)
      {
        if (ctx->n_dmsg == 0)
          {
            ctx->error = "the loopback group only carries the single-round ghost update and its reverse";
            return ASM_ERR_NOT_SUPPORTED;
          }
        return exchange_loopback(ctx, v, ctx->stream, add);
      }
    if (!add && ctx->n_dmsg > 0)
      {
        // one round: pack all boxes, one grouped send/recv with every neighbour, unpack all boxes
        BOX_TRY(vec_boxes(g, 0, v, ctx->dbuf[0], ctx->dsend, ctx->stream));
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < ctx->n_dmsg; ++i)
          if (ctx->dmsg[i].send_count > 0)
            NCCL_TRY(ncclSend(ctx->dbuf[0] + ctx->dmsg[i].send_offset, ctx->dmsg[i].send_count, ncclDouble, ctx->dmsg[i].peer,
                              ctx->comm, ctx->stream));
        for (int i = ctx->n_dmsg - 1; i >= 0; --i)
          if (ctx->dmsg[i].recv_count > 0)
            NCCL_TRY(ncclRecv(ctx->dbuf[1] + ctx->dmsg[i].recv_offset, ctx->dmsg[i].recv_count, ncclDouble, ctx->dmsg[i].peer,
                              ctx->comm, ctx->stream));
        NCCL_TRY(ncclGroupEnd());
        BOX_TRY(vec_boxes(g, 1, v, ctx->dbuf[1], ctx->drecv, ctx->stream));
        return ASM_OK;
      }
    static const bool dimension_ordered = std::getenv("ASM_COMPRESS_BY_DIMENSION") != nullptr; // experiments: the three-round form
    if (add && ctx->n_dmsg > 0 && !dimension_ordered)
      {
        // compress(add) in one round, the single-round ghost update run backwards: every ghost box goes straight
        // to the rank that owns it (face, edge and corner neighbours alike) and is added there; then the
        // ghost boxes are zeroed (zero_out_ghost_values).  Sends in array order, receives in reverse order:
        // the pairing argument of the update applies unchanged.
        BOX_TRY(vec_boxes(g, 0, v, ctx->dbuf[1], ctx->drecv, ctx->stream));
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < ctx->n_dmsg; ++i)
          if (ctx->dmsg[i].recv_count > 0)
            NCCL_TRY(ncclSend(ctx->dbuf[1] + ctx->dmsg[i].recv_offset, ctx->dmsg[i].recv_count, ncclDouble, ctx->dmsg[i].peer,
                              ctx->comm, ctx->stream));
        for (int i = ctx->n_dmsg - 1; i >= 0; --i)
          if (ctx->dmsg[i].send_count > 0)
            NCCL_TRY(ncclRecv(ctx->dbuf[0] + ctx->dmsg[i].send_offset, ctx->dmsg[i].send_count, ncclDouble, ctx->dmsg[i].peer,
                              ctx->comm, ctx->stream));
        NCCL_TRY(ncclGroupEnd());
        BOX_TRY(vec_boxes(g, 2, v, ctx->dbuf[0], ctx->dsend, ctx->stream));
        BOX_TRY(vec_boxes(g, 3, v, nullptr, ctx->drecv, ctx->stream));
        return ASM_OK;
      }
    asm_exchange_step steps[3];
    const int         n_steps = build_exchange_plan(ctx->s, add, steps);
    for (int i = 0; i < n_steps; ++i)
      {
        const asm_exchange_step &st = steps[i];
        long long                cnt_send[2], cnt_recv[2];
        for (int sd = 0; sd < 2; ++sd)
          {
            cnt_send[sd] = st.peer[sd] < 0 ? 0 : (long long)st.send_size[sd][0] * st.send_size[sd][1] * st.send_size[sd][2];
            cnt_recv[sd] = st.peer[sd] < 0 ? 0 : (long long)st.recv_size[sd][0] * st.recv_size[sd][1] * st.recv_size[sd][2];
            if (cnt_send[sd] > ctx->hbuf_size || cnt_recv[sd] > ctx->hbuf_size)
              {
                ctx->error = "internal error: halo buffer too small";
                return ASM_ERR_INVALID_VALUE;
              }
            if (cnt_send[sd] > 0)
              BOX_TRY(vec_box(g, 0, v, ctx->hbuf[sd], st.send_lo[sd], st.send_size[sd], ctx->stream));
          }
        NCCL_TRY(ncclGroupStart());
        // sends: low then high; receives: high then low, so that with two ranks in a
        // periodic direction (both neighbours are the same peer) the k-th send of one
        // side meets the k-th receive of the other
        for (int sd = 0; sd < 2; ++sd)
          if (cnt_send[sd] > 0)
            NCCL_TRY(ncclSend(ctx->hbuf[sd], cnt_send[sd], ncclDouble, st.peer[sd], ctx->comm, ctx->stream));
        for (int sd = 1; sd >= 0; --sd)
          if (cnt_recv[sd] > 0)
            NCCL_TRY(ncclRecv(ctx->hbuf[2 + sd], cnt_recv[sd], ncclDouble, st.peer[sd], ctx->comm, ctx->stream));
        NCCL_TRY(ncclGroupEnd());
        for (int sd = 0; sd < 2; ++sd)
          if (cnt_recv[sd] > 0)
            BOX_TRY(vec_box(g, add ? 2 : 1, v, ctx->hbuf[2 + sd], st.recv_lo[sd], st.recv_size[sd], ctx->stream));
        if (add) // zero_out_ghost_values
          for (int sd = 0; sd < 2; ++sd)
            if (cnt_send[sd] > 0)
              BOX_TRY(vec_box(g, 3, v, nullptr, st.send_lo[sd], st.send_size[sd], ctx->stream));
      }
    return ASM_OK;
  }

  // global sum of up to 8 device scalars (d_scalar[0 .. n)) -> host
  int
  reduce_scalar(asm_ctx *ctx, double *out, const int n = 1)
  {
    if (multi_rank(ctx))
      {
        if (ctx->loop)
          {
            if (n != 1)
              {
                ctx->error = "loopback group: one scalar per reduction";
                return ASM_ERR_NOT_SUPPORTED;
              }
            LoopbackGroup &lg = *ctx->loop;
            HIP_TRY(hipMemcpyAsync(ctx->h_scalar, ctx->d_scalar, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            lg.scalars[ctx->s.prm.rank] = *ctx->h_scalar;
            lg.barrier();
            double sum = 0.;
            for (const double x : lg.scalars)
              sum += x;
            lg.barrier();
            *out = sum;
            return ASM_OK;
          }
        if (ctx->comm == nullptr)
          {
            ctx->error = "world_size > 1 but asm_comm_init() has not been called";
            return ASM_ERR_INVALID_VALUE;
          }
        if (ctx->exchange_pending) // one RCCL operation of this communicator at a time
          {
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_exchanged, 0));
            ctx->exchange_pending = false;
          }
        NCCL_TRY(ncclAllReduce(ctx->d_scalar, ctx->d_scalar, n, ncclDouble, ncclSum, ctx->comm, ctx->stream));
      }
    HIP_TRY(hipMemcpyAsync(ctx->h_scalar, ctx->d_scalar, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i)
      out[i] = ctx->h_scalar[i];
    return ASM_OK;
  }

  int
  dot(asm_ctx *ctx, const double *x, const double *y, double *out)
  {
    HIP_TRY(hipMemsetAsync(ctx->d_scalar, 0, sizeof(double), ctx->stream));
    vec_dot(ctx->g, x, y, ctx->d_scalar, ctx->stream);
    return reduce_scalar(ctx, out);
  }

  // out[i] = <w, v[i]>, i < n: w is read once per 8 vectors, one host synchronisation per 8 inner products
  int
  multi_dot(asm_ctx *ctx, const double *w, const double *const *v, const int n, double *out)
  {
    if (ctx->loop)
      {
        for (int i = 0; i < n; ++i)
          if (const int err = dot(ctx, w, v[i], &out[i]))
            return err;
        return ASM_OK;
      }
    for (int c = 0; c < n; c += 8)
      {
        VecList L;
        L.n = std::min(8, n - c);
        for (int i = 0; i < 8; ++i)
          {
            L.v[i] = v[c + std::min(i, L.n - 1)];
            L.a[i] = 0.;
          }
        HIP_TRY(hipMemsetAsync(ctx->d_scalar, 0, 8 * sizeof(double), ctx->stream));
        vec_multi_dot(ctx->g, w, L, ctx->d_scalar, ctx->stream);
        if (const int err = reduce_scalar(ctx, out + c, L.n))
          return err;
      }
    return ASM_OK;
  }

  // w += sum_i a[i] v[i] (whole allocation), *norm2 = |w|^2 afterwards (owned DoFs, all ranks)
  int
  multi_axpy_norm(asm_ctx *ctx, double *w, const double *const *v, const double *a, const int n, double *norm2)
  {
    if (w == ctx->exchanged_vec)
      ctx->exchanged_vec = nullptr;
    HIP_TRY(hipMemsetAsync(ctx->d_scalar, 0, sizeof(double), ctx->stream));
    for (int c = 0; c < std::max(n, 1); c += 8)
      {
        VecList L;
        L.n = std::max(0, std::min(8, n - c));
        for (int i = 0; i < 8; ++i)
          {
            L.v[i] = L.n > 0 ? v[c + std::min(i, L.n - 1)] : w;
            L.a[i] = (i < L.n) ? a[c + i] : 0.;
          }
        const bool last = c + 8 >= n;
        vec_multi_axpy(ctx->g, w, L, ctx->s.n_alloc, last ? ctx->d_scalar : nullptr, ctx->stream);
      }
    return reduce_scalar(ctx, norm2);
  }

  // testing aid: the single-round ghost update through the loopback group (host-synchronous)
  int
  exchange_loopback(asm_ctx *ctx, double *v, hipStream_t st, const bool add)
  {
    LoopbackGroup &lg = *ctx->loop;
    if (add)
      {
        // compress(add): the single-round update run backwards (see exchange): ghost boxes -> their owners, added there
        BOX_TRY(vec_boxes(ctx->g, 0, v, ctx->dbuf[1], ctx->drecv, st));
        HIP_TRY(hipStreamSynchronize(st));
        lg.barrier(); // every rank has packed its ghost boxes
        for (int i = 0; i < ctx->n_dmsg; ++i)
          {
            const asm_direct_message &m = ctx->dmsg[i];
            if (m.send_count == 0)
              continue;
            const asm_ctx *           pc = lg.ranks[m.peer];
            const asm_direct_message *pm = nullptr;
            for (int j = 0; j < pc->n_dmsg; ++j)
              if (pc->dmsg[j].peer == ctx->s.prm.rank && pc->dmsg[j].offset[0] == -m.offset[0] && pc->dmsg[j].offset[1] == -m.offset[1] &&
                  pc->dmsg[j].offset[2] == -m.offset[2])
                pm = &pc->dmsg[j];
            if (pm == nullptr || pm->recv_count != m.send_count)
              {
                ctx->error = "loopback exchange: send and receive boxes do not match";
                return ASM_ERR_INVALID_VALUE;
              }
            HIP_TRY(hipMemcpyAsync(ctx->dbuf[0] + m.send_offset, pc->dbuf[1] + pm->recv_offset, m.send_count * sizeof(double),
                                   hipMemcpyDeviceToDevice, st));
          }
        HIP_TRY(hipStreamSynchronize(st));
        lg.barrier(); // every rank has read its neighbours' buffers
        BOX_TRY(vec_boxes(ctx->g, 2, v, ctx->dbuf[0], ctx->dsend, st));
        BOX_TRY(vec_boxes(ctx->g, 3, v, nullptr, ctx->drecv, st));
        return ASM_OK;
      }
    BOX_TRY(vec_boxes(ctx->g, 0, v, ctx->dbuf[0], ctx->dsend, st));
    HIP_TRY(hipStreamSynchronize(st));
    lg.barrier(); // every rank has packed
    for (int i = 0; i < ctx->n_dmsg; ++i)
      {
        const asm_direct_message &m = ctx->dmsg[i];
        if (m.recv_count == 0)
          continue;
        const asm_ctx *           pc = lg.ranks[m.peer];
        const asm_direct_message *pm = nullptr;
        for (int j = 0; j < pc->n_dmsg; ++j)
          if (pc->dmsg[j].peer == ctx->s.prm.rank && pc->dmsg[j].offset[0] == -m.offset[0] && pc->dmsg[j].offset[1] == -m.offset[1] &&
              pc->dmsg[j].offset[2] == -m.offset[2])
            pm = &pc->dmsg[j];
        if (pm == nullptr || pm->send_count != m.recv_count)
          {
            ctx->error = "loopback exchange: send and receive boxes do not match";
            return ASM_ERR_INVALID_VALUE;
          }
        HIP_TRY(hipMemcpyAsync(ctx->dbuf[1] + m.recv_offset, pc->dbuf[0] + pm->send_offset, m.recv_count * sizeof(double),
                               hipMemcpyDeviceToDevice, st));
      }
    HIP_TRY(hipStreamSynchronize(st));
    lg.barrier(); // every rank has read its neighbours' send buffers
    BOX_TRY(vec_boxes(ctx->g, 1, v, ctx->dbuf[1], ctx->drecv, st));
    return ASM_OK;
  }

  // the single-round ghost update of v on comm_stream, started when ev_boundary has fired
  int
  exchange_overlapped(asm_ctx *ctx, double *v)
  {
    HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_boundary, 0));
    if (ctx->loop)
      {
        const int err = exchange_loopback(ctx, v, ctx->comm_stream);
        if (err)
          return err;
        HIP_TRY(hipEventRecord(ctx->ev_exchanged, ctx->comm_stream));
        ctx->exchanged_vec    = v;
        ctx->exchange_pending = true;
        return ASM_OK;
      }
    BOX_TRY(vec_boxes(ctx->g, 0, v, ctx->dbuf[0], ctx->dsend, ctx->comm_stream));
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < ctx->n_dmsg; ++i)
      if (ctx->dmsg[i].send_count > 0)
        NCCL_TRY(ncclSend(ctx->dbuf[0] + ctx->dmsg[i].send_offset, ctx->dmsg[i].send_count, ncclDouble, ctx->dmsg[i].peer, ctx->comm,
                          ctx->comm_stream));
    for (int i = ctx->n_dmsg - 1; i >= 0; --i)
      if (ctx->dmsg[i].recv_count > 0)
        NCCL_TRY(ncclRecv(ctx->dbuf[1] + ctx->dmsg[i].recv_offset, ctx->dmsg[i].recv_count, ncclDouble, ctx->dmsg[i].peer, ctx->comm,
                          ctx->comm_stream));
    NCCL_TRY(ncclGroupEnd());
    BOX_TRY(vec_boxes(ctx->g, 1, v, ctx->dbuf[1], ctx->drecv, ctx->comm_stream));
    HIP_TRY(hipEventRecord(ctx->ev_exchanged, ctx->comm_stream));
    ctx->exchanged_vec    = v;
    ctx->exchange_pending = true;
    return ASM_OK;
  }

  // The wave-pair sweep where it applies (single-phase launches); false: not applicable, nothing launched
  bool
  pair_laplace(asm_ctx *ctx, const TiledArgs &t, hipStream_t st)
  {
    // (the constrained DoFs - 0, or r_c = b_c - x_c - are set by the last workgroups of the same launch)
    return ctx->pair_a && pair_sweep(ctx->g, ctx->kron_a, t, st);
  }
  bool
  pair_fdm(asm_ctx *ctx, const TiledArgs &t, hipStream_t st)
  {
    // (the constrained DoFs - unit rows of P^-1 - are set by the last workgroups of the same launch)
    return ctx->pair_p && pair_sweep(ctx->g, ctx->kron_p, t, st);
  }

  // launch a tiled sweep; inside cheb_run with several ranks: boundary workgroups, exchange of the
  // result beside the interior workgroups (unless nobody reads the result's ghosts afterwards)
  template <typename Launch>
  int
  launch_tiled_sweep(asm_ctx *ctx, TiledArgs &a, const Launch &launch, bool *ok)
  {
    if (a.dst == ctx->exchanged_vec)
      ctx->exchanged_vec = nullptr; // overwritten: its ghosts are no longer current
    const bool ov = ctx->overlap && ctx->in_cheb && !ctx->external_exchange && !(ctx->last_sweep && !ctx->keep_ghosts);
    for (int d = 0; d < 3; ++d)
      for (int f = 0; f < 2; ++f)
        a.iface[d][f] = ctx->s.face[d][f] == FACE_INTERFACE;
    a.device = ctx->device;
    a.n_cu   = ctx->n_cu;
    a.rows   = ctx->laplace_rows;
    a.order  = (a.mode == OP_PLAIN || a.mode == OP_RESIDUAL) ? ctx->order_a : ctx->order_p;
    a.pair_order = ctx->order_pair;
    a.pair_phase_order = ctx->order_pair_phase;
    a.pair_n_first     = ctx->pair_n_first;
    a.pair_zchunks = ctx->pair_zchunks;
    a.pair_waves   = ctx->pair_waves;
    if (!ov && ctx->debug_two_phase)
      {
        a.phase = 1;
        *ok     = launch(a, ctx->stream);
        a.phase = 2;
        *ok     = *ok && launch(a, ctx->stream);
        return ASM_OK;
      }
    if (!ov)
      {
        a.phase = 0;
        *ok     = launch(a, ctx->stream);
        return ASM_OK;
      }
    if (ctx->bnd_stream == nullptr)
      {
        // the boundary workgroups (padded to whole rounds of compute units), then the others, on one stream
        a.phase = 1;
        *ok     = launch(a, ctx->stream);
        if (!*ok)
          return ASM_OK;
        HIP_TRY(hipEventRecord(ctx->ev_boundary, ctx->stream));
        a.phase = 2;
        *ok     = launch(a, ctx->stream);
        // enqueued behind the interior launch on the host, beside it on the device
        return exchange_overlapped(ctx, a.dst);
      }
    // Boundary workgroups on a high-priority stream, the interior ones beside them on the main stream: the boundary
    // launch does not leave compute units idle while its last workgroups finish, and the exchange starts as soon as
    // the boundary workgroups (dispatched first) are through.
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->bnd_stream, ctx->ev_start, 0));
    a.phase = 1;
    *ok     = launch(a, ctx->bnd_stream);
    if (!*ok)
      return ASM_OK;
    HIP_TRY(hipEventRecord(ctx->ev_boundary, ctx->bnd_stream));
    a.phase = 2;
    *ok     = launch(a, ctx->stream);
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_boundary, 0)); // the sweep is complete on the main stream
    return exchange_overlapped(ctx, a.dst);
  }

  // ---- sweeps --------------------------------------------------------------
  // dst = A src (OP_PLAIN) or dst = b - A src with identity rows on constrained
  // DoFs (OP_RESIDUAL); src's ghosts are updated here.
  int
  op_apply_impl(asm_ctx *ctx, const int mode, double *dst, double *src, const double *b)
  {
    int err = exchange(ctx, src, false);
    if (err)
      return err;
    if (ctx->s.laplace_tiled)
      {
        // the tiled Laplace kernel covers Cartesian geometry; deformed cells take the
        // generic (quadrature-point Jacobian) kernel below
        TiledArgs a{mode, src, dst, b, nullptr, nullptr, 0., 0.};
        prof_mark(ctx, 0, true);
        bool ok = false;
        err     = launch_tiled_sweep(
          ctx, a, [&](const TiledArgs &t, hipStream_t st) { return pair_laplace(ctx, t, st) || laplace_tiled(ctx->g, ctx->op, t, st); }, &ok);
        if (err)
          return err;
        if (ok)
          {
            prof_mark(ctx, 0, false);
            return ASM_OK;
          }
        // (the begin event is simply re-recorded by the generic launch below)
      }
    if (dst == ctx->exchanged_vec)
      ctx->exchanged_vec = nullptr;
    // one rank: the residual is accumulated straight into a copy of b (no zeroing pass, no separate b - A x pass)
    const bool fused_residual = mode == OP_RESIDUAL && !multi_rank(ctx);
    // (small levels are bound by the number of launches: the copy and the Dirichlet planes in one)
    const bool init_kernel = fused_residual && ctx->s.n_alloc < (1ll << 22) && dst != b && dst != src;
    if (init_kernel)
      vec_residual_init(ctx->g, dst, b, src, ctx->stream);
    else if (fused_residual)
      HIP_TRY(hipMemcpyAsync(dst, b, ctx->s.n_alloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    else
      HIP_TRY(hipMemsetAsync(dst, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    prof_mark(ctx, 0, true);
    const bool ok = laplace_generic(ctx->g, ctx->op, src, dst, false, ctx->stream, fused_residual ? -1.0 : 1.0);
    prof_mark(ctx, 0, false);
    if (!ok)
      {
        ctx->error = "generic Laplace kernel not available for this degree";
        return ASM_ERR_NOT_SUPPORTED;
      }
    err = exchange(ctx, dst, true);
    if (err)
      return err;
    if (mode == OP_RESIDUAL)
      {
        if (!fused_residual)
          vec_residual(dst, b, ctx->s.n_alloc, ctx->stream);
        if (!init_kernel)
          vec_constrained(ctx->g, 2, dst, src, b, ctx->stream); // r_c = b_c - x_c
      }
    return ASM_OK;
  }

  int
  op_apply(asm_ctx *ctx, const int mode, double *dst, double *src, const double *b)
  {
    const int err = op_apply_impl(ctx, mode, dst, src, b);
    return err ? err : launch_status(ctx, "the Laplace sweep");
  }

  // dst = [f(x, x_old) +] f2 * P^-1 src, see TiledMode
  int
  pc_apply_impl(asm_ctx *ctx, const int mode, double *dst, double *src, const double *x, const double *x_old, const double f1,
                const double f2)
  {
    const long long n = ctx->s.n_alloc;
    if (ctx->s.prm.precond_type == ASM_PRECOND_DIAGONAL)
      {
        // DiagonalMatrixPrePost::vmult, include/preconditioners.h:965-993
        double *t = (mode == PC_PLAIN) ? dst : ctx->T;
        vec_mul(t, src, ctx->inv_diag, n, ctx->stream);
        if (mode == PC_FIRST)
          vec_scale(dst, t, f2, n, ctx->stream);
        else if (mode == PC_STEP1)
          {
            vec_scale(dst, x, f1, n, ctx->stream);
            vec_axpby(dst, f2, t, 1., n, ctx->stream);
          }
        else if (mode == PC_UPDATE)
          vec_cheb_update(dst, x, x_old, t, f1, f2, n, ctx->stream);
        return ASM_OK;
      }
    int err = exchange(ctx, src, false);
    if (err)
      return err;
    if (ctx->s.fdm_tiled)
      {
        TiledArgs a{mode, src, dst, nullptr, x, x_old, f1, f2};
        prof_mark(ctx, 1, true);
        bool ok = false;
        err     = launch_tiled_sweep(
          ctx, a, [&](const TiledArgs &t, hipStream_t st) { return pair_fdm(ctx, t, st) || fdm_tiled(ctx->g, ctx->f, t, st); }, &ok);
        if (err)
          return err;
        prof_mark(ctx, 1, false);
        if (!ok)
          {
            ctx->error = "tiled FDM kernel not available for this configuration";
            return ASM_ERR_NOT_SUPPORTED;
          }
        return ASM_OK;
      }
    if (dst == ctx->exchanged_vec)
      ctx->exchanged_vec = nullptr;
    FdmFused fu;
    if (ctx->f.s7_flag[0] != nullptr && !multi_rank(ctx))
      {
        // 7^3 patches, one rank: patch results summed in LDS along z, the vector update applied with the store
        fu.mode  = mode;
        fu.x     = x;
        fu.x_old = x_old;
        fu.f1    = f1;
        fu.f2    = f2;
        prof_mark(ctx, 1, true);
        const bool ok = fdm_slice7_acc(ctx->g, ctx->f, src, dst, fu, ctx->stream);
        prof_mark(ctx, 1, false);
        if (!ok)
          {
            ctx->error = "accumulating 7^3 FDM kernel not available for this configuration";
            return ASM_ERR_NOT_SUPPORTED;
          }
        return ASM_OK;
      }
    if (ctx->patch_tmp != nullptr && !multi_rank(ctx))
      {
        // one rank: the kernel that sums the patch results per DoF applies the vector update as well
        // (no zeroing, no atomics, no separate update passes)
        fu.mode  = mode;
        fu.x     = x;
        fu.x_old = x_old;
        fu.f1    = f1;
        fu.f2    = f2;
        prof_mark(ctx, 1, true);
        const bool ok = fdm_generic(ctx->g, ctx->f, src, dst, ctx->patch_tmp, fu, ctx->stream);
        prof_mark(ctx, 1, false);
        if (!ok)
          {
            ctx->error = "generic FDM kernel not available for this patch size";
            return ASM_ERR_NOT_SUPPORTED;
          }
        return ASM_OK;
      }
    if (ctx->patch_tmp == nullptr && !multi_rank(ctx) && ctx->s.prm.weighting_type == ASM_W_RAS && ctx->s.prm.element_centric != 0 &&
        ctx->ras_direct)
      {
        // restricted additive Schwarz on element-centred patches, one rank: every DoF is written by exactly one patch,
        // which stores the updated value itself (no zeroing, no atomics, no update passes)
        fu.mode   = mode;
        fu.x      = x;
        fu.x_old  = x_old;
        fu.f1     = f1;
        fu.f2     = f2;
        fu.direct = 1;
        prof_mark(ctx, 1, true);
        const bool ok = fdm_generic(ctx->g, ctx->f, src, dst, nullptr, fu, ctx->stream);
        prof_mark(ctx, 1, false);
        if (!ok)
          {
            ctx->error = "generic FDM kernel not available for this patch size";
            return ASM_ERR_NOT_SUPPORTED;
          }
        vec_constrained_fused(ctx->g, fu, dst, src, ctx->stream);
        return ASM_OK;
      }
    double *t = (mode == PC_PLAIN) ? dst : ctx->T;
    if (ctx->patch_tmp == nullptr)
      HIP_TRY(hipMemsetAsync(t, 0, n * sizeof(double), ctx->stream));
    prof_mark(ctx, 1, true);
    const bool ok = fdm_generic(ctx->g, ctx->f, src, t, ctx->patch_tmp, fu, ctx->stream);
    prof_mark(ctx, 1, false);
    if (!ok)
      {
        ctx->error = "generic FDM kernel not available for this patch size";
        return ASM_ERR_NOT_SUPPORTED;
      }
    // RAS could skip this exchange in the reference (needs_compression, matrix_free.h:654-668);
    // here the patch owning a shared plane may sit on the lower rank, so it stays
    err = exchange(ctx, t, true);
    if (err)
      return err;
    if (mode != PC_PLAIN)
      vec_constrained(ctx->g, 1, t, src, nullptr, ctx->stream); // unit rows on constrained DoFs
    if (mode == PC_FIRST)
      vec_scale(dst, t, f2, n, ctx->stream);
    else if (mode == PC_STEP1) // dst = f1 * x + f2 * t   (f1 carries 1 + factor1)
      {
        vec_scale(dst, x, f1, n, ctx->stream);
        vec_axpby(dst, f2, t, 1., n, ctx->stream);
      }
    else if (mode == PC_UPDATE)
      vec_cheb_update(dst, x, x_old, t, f1, f2, n, ctx->stream);
    return ASM_OK;
  }

  int
  pc_apply(asm_ctx *ctx, const int mode, double *dst, double *src, const double *x, const double *x_old, const double f1,
           const double f2)
  {
    const int err = pc_apply_impl(ctx, mode, dst, src, x, x_old, f1, f2);
    return err ? err : launch_status(ctx, "the preconditioner sweep");
  }

  // PreconditionChebyshev::vmult / step [deal.II], first-kind polynomial:
  //   x_1     = x_0 + 1/theta P^-1 (b - A x_0)
  //   x_{j+1} = x_j + rho_j rho_{j-1} (x_j - x_{j-1}) + 2 rho_j/delta P^-1 (b - A x_j)
  int
  cheb_run(asm_ctx *ctx, double *dst, const double *b, const bool zero_start)
  {
    if (!ctx->ev_set)
      {
        ctx->error = "Chebyshev eigenvalues not set: call asm_cheb_estimate_eigenvalues or asm_cheb_set_eigenvalue";
        return ASM_ERR_INVALID_VALUE;
      }
    const int    degree = ctx->s.prm.cheb_degree;
    const double theta = ctx->theta, delta = ctx->delta;
    const bool   fourth  = ctx->s.prm.polynomial_type == ASM_POLY_FOURTH_KIND;
    // factor of the first update: 1/theta (1st kind), 4/(3 lambda_max) (4th kind; delta holds lambda_max)
    const double first   = fourth ? 4. / (3. * delta) : 1. / theta;
    const bool   iterate = degree >= 2 && std::fabs(delta) >= 1e-40;
    const int    k_total = iterate ? degree : 1;
    double *     A = dst, *S = ctx->S, *R = ctx->R;
    double *     cur, *prev; // cur holds x_j, prev holds x_{j-1}
    int          err;
    // sweeps of this call may exchange their results beside their own interior work
    struct Scope
    {
      asm_ctx *c;
      explicit Scope(asm_ctx *ctx)
        : c(ctx)
      {
        c->in_cheb    = true;
        c->last_sweep = false;
        if (!c->keep_ghosts)
          c->exchanged_vec = nullptr;
      }
      ~Scope()
      {
        c->in_cheb    = false;
        c->last_sweep = false;
      }
    } scope(ctx);
    if (zero_start)
      {
        // the final iterate lands in A if the first one goes to A for odd k, to S for even k
        cur             = (k_total % 2) ? A : S;
        ctx->last_sweep = k_total == 1;
        err             = pc_apply(ctx, PC_FIRST, cur, const_cast<double *>(b), nullptr, nullptr, 0., first);
        if (err)
          return err;
        prev = nullptr;
      }
    else
      {
        err = op_apply(ctx, OP_RESIDUAL, R, A, b);
        if (err)
          return err;
        if (k_total == 1)
          {
            ctx->last_sweep = true;
            return pc_apply(ctx, PC_STEP1, A, R, A, nullptr, 1., first);
          }
        err = pc_apply(ctx, PC_STEP1, S, R, A, nullptr, 1., first);
        if (err)
          return err;
        cur  = S;
        prev = A;
      }
    double rhok = delta / theta;
    const double sigma = theta / delta;
    for (int k = 1; k < k_total; ++k)
      {
        double f1, f2;
        if (fourth)
          {
            // x_{j+1} = x_j + (2j-1)/(2j+3) (x_j - x_{j-1}) + (8j+4)/(2j+3) / lambda_max P^-1 r_j, j = k (Lottes 2022;
            // deal.II PreconditionChebyshev, PolynomialType::fourth_kind)
            f1 = (2. * k - 1.) / (2. * k + 3.);
            f2 = (8. * k + 4.) / (delta * (2. * k + 3.));
          }
        else
          {
            const double rhokp = 1. / (2. * sigma - rhok);
            f1                 = rhokp * rhok;
            f2                 = 2. * rhokp / delta;
            rhok               = rhokp;
          }
        err  = op_apply(ctx, OP_RESIDUAL, R, cur, b);
        if (err)
          return err;
        ctx->last_sweep = k == k_total - 1;
        if (prev == nullptr)
          {
            // x_0 = 0: x_2 = (1 + f1) x_1 + f2 t
            double *out = (cur == A) ? S : A;
            err         = pc_apply(ctx, PC_STEP1, out, R, cur, nullptr, 1. + f1, f2);
            prev        = cur;
            cur         = out;
          }
        else
          {
            err = pc_apply(ctx, PC_UPDATE, prev, R, cur, prev, f1, f2);
            std::swap(cur, prev);
          }
        if (err)
          return err;
      }
    if (cur != A)
      {
        if (ctx->exchange_pending) // the copy includes the ghost planes
          {
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_exchanged, 0));
            ctx->exchange_pending = false;
          }
        HIP_TRY(hipMemcpyAsync(A, cur, ctx->s.n_alloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        if (ctx->exchanged_vec == cur)
          ctx->exchanged_vec = A;
        else if (ctx->exchanged_vec == A)
          ctx->exchanged_vec = nullptr;
      }
    return ASM_OK;
  }

  // The tiled Laplace sweep exists with 8 (first generation), 9, 10 and 11 cell rows per workgroup.
  // Which is fastest depends on how the tiles fit the block and on the device (halo rows, tile
  // quantisation, rounds of workgroups): measured once per context, two launches per candidate on the
  // scratch vectors.  All candidates sum the contributions to a DoF in the same order: the choice
  // does not change results.  ASM_TILED_ROWS=0|9|10|11 fixes it (experiments).
  std::mutex                         g_tune_mutex;
  std::map<std::array<int, 15>, int> g_tune_cache;
  int
  tune_tiled_laplace(asm_ctx *ctx)
  {
    if (!ctx->s.laplace_tiled)
      return ASM_OK;
    if (const char *env = std::getenv("ASM_TILED_ROWS"))
      {
        const int r = std::atoi(env);
        ctx->laplace_rows = (r >= 9 && r <= 11) ? r : 0;
        return ASM_OK;
      }
    // one measurement per device, block shape and boundary kinds in a process: contexts of the same shape (the ranks'
    // blocks in a loopback group, repeated creation) take the same variant, and a creation beside other work on the GPU
    // does not measure again (ASM_TILED_ROWS still overrides)
    std::array<int, 15> key;
    key[0] = ctx->device;
    for (int d = 0; d < 3; ++d)
      {
        key[1 + d]  = ctx->g.nc[d];
        key[4 + d]  = ctx->g.wrap[d];
        key[7 + d]  = ctx->g.dir_lo[d] * 2 + ctx->g.dir_hi[d];
        key[10 + d] = ctx->g.ext_hi[d] - ctx->g.ext_lo[d];
      }
    key[13] = ctx->op.cartesian;
    key[14] = ctx->n_cu;
    // (overrides of the tiling change what is measured: such contexts measure for themselves)
    const bool use_cache = std::getenv("ASM_TILED_ZCHUNKS") == nullptr && std::getenv("ASM_TILED_TAIL") == nullptr;
    if (use_cache)
      {
        std::lock_guard<std::mutex> lock(g_tune_mutex);
        const auto                  hit = g_tune_cache.find(key);
        if (hit != g_tune_cache.end())
          {
            ctx->laplace_rows = hit->second;
            return ASM_OK;
          }
      }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int  cand[4] = {0, 9, 10, 11};
    float      ms_of[4];
    const int *order_of[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i)
      {
        // every candidate with its own work order (kernels.h: tiled_order), as it would run
        if (std::getenv("ASM_NO_TILED_ORDER") == nullptr)
          {
            std::vector<int> order;
            tiled_order(ctx->g, ctx->n_cu, cand[i], order);
            if (const int err = upload(ctx, order, &order_of[i]))
              {
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                return err;
              }
          }
        // the residual form the Chebyshev iteration uses: dst = T, src = R, b = S (zero vectors)
        TiledArgs a{OP_RESIDUAL, ctx->R, ctx->T, ctx->S, nullptr, nullptr, 0., 0.};
        a.device = ctx->device;
        a.n_cu   = ctx->n_cu;
        a.rows   = cand[i];
        a.order  = order_of[i];
        bool ok  = true;
        for (int rep = 0; rep < 2 && ok; ++rep) // warm-up (code object, caches, clocks)
          ok = laplace_tiled(ctx->g, ctx->op, a, ctx->stream);
        (void)hipEventRecord(e0, ctx->stream);
        for (int rep = 0; rep < 4 && ok; ++rep)
          ok = laplace_tiled(ctx->g, ctx->op, a, ctx->stream);
        (void)hipEventRecord(e1, ctx->stream);
        float ms = 1e30f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
          ok = false;
        ms_of[i] = ok ? ms : 1e30f;
      }
    // the first-generation kernel unless another one is clearly (2 %) faster
    int pick = 0, pick_i = 0;
    float best = ms_of[0];
    for (int i = 1; i < 4; ++i)
      if (ms_of[i] < 0.98f * ms_of[0] && ms_of[i] < best)
        {
          best   = ms_of[i];
          pick   = cand[i];
          pick_i = i;
        }
    ctx->order_a = order_of[pick_i];
    ctx->order_p = order_of[0]; // (the FDM sweep is the first-generation kernel: same tiling as candidate 0)
    for (int i = 1; i < 4; ++i)
      if (i != pick_i)
        release(ctx, order_of[i]); // (every launch above has completed: hipEventSynchronize)
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->laplace_rows = pick;
    if (use_cache)
      {
        std::lock_guard<std::mutex> lock(g_tune_mutex);
        g_tune_cache[key] = pick;
      }
    HIP_TRY(hipMemsetAsync(ctx->T, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    return ASM_OK;
  }

  // Wave-pair or tiled kernel, per sweep: measured once per device, block shape and boundary kinds in a process (the
  // faster one depends on how the tiles of either family fit the block); ASM_PAIR=1 / 0 takes the pair kernels wherever
  // they apply / nowhere.  Called when everything else of the context stands (tables, work orders, scratch vectors).
  std::map<std::array<int, 15>, int> g_pair_cache; // bit 0: A, bit 1: P, bit 2: two waves per tile, from bit 3: z chunks
  int
  tune_pair(asm_ctx *ctx)
  {
    ctx->pair_a = ctx->pair_p = false;
    if (!ctx->kron_a.valid && !ctx->kron_p.valid)
      return ASM_OK;
    // ASM_PAIR = 0: tiled kernels, 1: pair kernels for both sweeps, 2: for P^-1 only, 3: for A only - no measurement
    // (ASM_PAIR_ZCHUNKS, ASM_PAIR_WAVES pin the rest: profiles of exactly the kernels a context had picked)
    if (const char *env = std::getenv("ASM_PAIR_WAVES"))
      ctx->pair_waves = std::atoi(env) == 2 ? 2 : 0;
    if (const char *env = std::getenv("ASM_PAIR"))
      {
        const int v = std::atoi(env);
        ctx->pair_a = ctx->kron_a.valid && (v == 1 || v == 3);
        ctx->pair_p = ctx->kron_p.valid && (v == 1 || v == 2);
        return ASM_OK;
      }
    // the number of z chunks of the pair sweeps: 0 = the schedule model's choice (whose work order the context holds)
    const auto set_chunks = [&](const int zchunks) -> int {
      if (zchunks == ctx->pair_zchunks)
        return ASM_OK;
      ctx->pair_zchunks = zchunks;
      return upload_pair_orders(ctx, zchunks); // (the launches with the tables of the chunk count before have completed)
    };
    std::array<int, 15> key;
    key[0] = ctx->device;
    for (int d = 0; d < 3; ++d)
      {
        key[1 + d]  = ctx->g.nc[d];
        key[4 + d]  = ctx->g.wrap[d];
        key[7 + d]  = ctx->g.dir_lo[d] * 2 + ctx->g.dir_hi[d];
        key[10 + d] = ctx->g.ext_hi[d] - ctx->g.ext_lo[d];
      }
    key[13] = (ctx->kron_a.valid ? 1 : 0) | (ctx->kron_p.valid ? 2 : 0) | (ctx->s.prm.weighting_type << 2);
    key[14] = ctx->n_cu;
    {
      std::lock_guard<std::mutex> lock(g_tune_mutex);
      const auto                  hit = g_pair_cache.find(key);
      if (hit != g_pair_cache.end())
        {
          ctx->pair_a = (hit->second & 1) != 0;
          ctx->pair_p     = (hit->second & 2) != 0;
          ctx->pair_waves = (hit->second & 4) != 0 ? 2 : 0;
          return set_chunks(hit->second >> 3);
        }
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    // operands that are not all zero (the clock the chip holds, and with it which family is faster, depends on the data)
    {
      const long long gb[3] = {0, 0, 0}, gs[3] = {1 << 20, 1 << 20, 1 << 20};
      vec_initial_guess(ctx->g, ctx->S, gb, gs, ctx->stream);
      vec_initial_guess(ctx->g, ctx->T, gb, gs, ctx->stream);
    }
    // what is timed is the Chebyshev step the sweeps are used in, with its operands (cheb_apply: R = b - A x, then
    // x += f P^-1 R in place; here x = S, b = T): which kernel is faster depends on what the sweep before it has left in
    // the caches.  A family without a tiled kernel is left out of the step (and keeps its generic kernel).
    // (a context with another preconditioner - the diagonal one of a multigrid level - has no FDM tables)
    const bool have_a = ctx->s.laplace_tiled, have_p = ctx->s.fdm_tiled && ctx->s.prm.precond_type == ASM_PRECOND_FDM;
    const auto time_step = [&](const bool pair_a, const bool pair_p) -> float {
      TiledArgs aa{OP_RESIDUAL, ctx->S, ctx->R, ctx->T, nullptr, nullptr, 0., 0.};
      TiledArgs ap{PC_STEP1, ctx->R, ctx->S, nullptr, ctx->S, nullptr, 1., 0.05};
      for (TiledArgs *a : {&aa, &ap})
        {
          a->device       = ctx->device;
          a->n_cu         = ctx->n_cu;
          a->rows         = ctx->laplace_rows;
          a->order        = a == &ap ? ctx->order_p : ctx->order_a;
          a->pair_order   = ctx->order_pair;
          a->pair_phase_order = ctx->order_pair_phase;
          a->pair_n_first     = ctx->pair_n_first;
          a->pair_zchunks = ctx->pair_zchunks;
          a->pair_waves   = ctx->pair_waves;
          for (int d = 0; d < 3; ++d)
            for (int f = 0; f < 2; ++f)
              a->iface[d][f] = ctx->s.face[d][f] == FACE_INTERFACE;
        }
      const bool was_a = ctx->pair_a, was_p = ctx->pair_p;
      ctx->pair_a = ctx->pair_p = true;
      // (a block with rank interfaces runs its sweeps inside the Chebyshev iteration in two phases, the ghost exchange
      //  beside the second one: the families are compared in that form - the tiled kernels then chunk differently)
      // (asm_comm_init switches the overlapped exchange on for a tiled context with neighbours unless ASM_NO_OVERLAP)
      const bool two_phase = multi_rank(ctx) && std::getenv("ASM_NO_OVERLAP") == nullptr;
      const auto sweep     = [&](TiledArgs &a, const bool fdm, const bool pair) {
        bool ok = true;
        for (int phase = two_phase ? 1 : 0; phase <= (two_phase ? 2 : 0) && ok; ++phase)
          {
            a.phase = phase;
            ok      = fdm ? (pair ? pair_fdm(ctx, a, ctx->stream) : fdm_tiled(ctx->g, ctx->f, a, ctx->stream)) :
                            (pair ? pair_laplace(ctx, a, ctx->stream) : laplace_tiled(ctx->g, ctx->op, a, ctx->stream));
          }
        return ok;
      };
      const auto step = [&]() {
        bool ok = true;
        if (have_a)
          ok = sweep(aa, false, pair_a);
        if (ok && have_p)
          ok = sweep(ap, true, pair_p);
        return ok;
      };
      bool ok = true;
      for (int rep = 0; rep < 2 && ok; ++rep)
        ok = step();
      (void)hipEventRecord(e0, ctx->stream);
      for (int rep = 0; rep < 5 && ok; ++rep)
        ok = step();
      (void)hipEventRecord(e1, ctx->stream);
      ctx->pair_a = was_a;
      ctx->pair_p = was_p;
      float ms = 1e30f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
        ok = false;
      return ok ? ms : 1e30f;
    };
    const bool can_pair_a = ctx->kron_a.valid && ctx->s.laplace_tiled, can_pair_p = ctx->kron_p.valid && have_p;
    // the number of z chunks: the schedule model (kernels_pair.hip: schedule) ranks the candidates, but its first ones
    // differ by less than its error, so they are measured - with the pair kernel for P^-1 where there is one (its main
    // use), otherwise for A
    int        err   = ASM_OK;
    const bool for_p = can_pair_p;
    if (std::getenv("ASM_PAIR_ZCHUNKS") == nullptr && (can_pair_a || can_pair_p))
      {
        // ... for both forms of the kernel (four and two waves per tile) unless ASM_PAIR_WAVES names one
        std::vector<int> cand;
        pair_chunk_candidates(ctx->g, ctx->n_cu, cand);
        const char *                                       env_w = std::getenv("ASM_PAIR_WAVES");
        std::vector<std::pair<float, std::pair<int, int>>> timed;
        for (const int waves : {4, 2})
          if (env_w == nullptr || std::atoi(env_w) == waves)
            for (std::size_t i = 0; i < cand.size() && i < 16 && !err; ++i)
              if (!(err = set_chunks(cand[i])))
                {
                  ctx->pair_waves = waves;
                  timed.emplace_back(time_step(!for_p, for_p), std::make_pair(cand[i], waves));
                }
        // (the four fastest again, three interleaved rounds, the best of those for each: a box's clock drifts within the
        //  measurement, and a first pass's lucky outlier must not decide)
        std::sort(timed.begin(), timed.end());
        const std::size_t  n_final = std::min<std::size_t>(timed.size(), 4);
        std::vector<float> again(n_final, 1e30f);
        for (int round = 0; round < 3 && !err; ++round)
          for (std::size_t i = 0; i < n_final && !err; ++i)
            if (!(err = set_chunks(timed[i].second.first)))
              {
                ctx->pair_waves = timed[i].second.second;
                again[i]        = std::min(again[i], time_step(!for_p, for_p));
              }
        int   best = 0, best_waves = 0;
        float best_time = 1e30f;
        for (std::size_t i = 0; i < n_final; ++i)
          if (again[i] < best_time)
            {
              best_time  = again[i];
              best       = timed[i].second.first;
              best_waves = timed[i].second.second;
            }
        ctx->pair_waves = best_waves;
        if (!err)
          err = set_chunks(best);
      }
    if (err)
      {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return err;
      }
    // (two interleaved rounds per choice, the better of each)
    const auto faster = [&](const bool pa1, const bool pp1, const bool pa0, const bool pp0) {
      float t1 = 1e30f, t0 = 1e30f;
      for (int round = 0; round < 2; ++round)
        {
          t1 = std::min(t1, time_step(pa1, pp1));
          t0 = std::min(t0, time_step(pa0, pp0));
        }
      return t1 < 0.985f * t0;
    };
    int pick = 0;
    if (can_pair_p)
      pick |= faster(false, true, false, false) ? 2 : 0;
    if (can_pair_a)
      pick |= faster(true, (pick & 2) != 0, false, (pick & 2) != 0) ? 1 : 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ctx->pair_a = (pick & 1) != 0;
    ctx->pair_p = (pick & 2) != 0;
    {
      std::lock_guard<std::mutex> lock(g_tune_mutex);
      g_pair_cache[key] = pick | (ctx->pair_waves == 2 ? 4 : 0) | (ctx->pair_zchunks << 3);
    }
    HIP_TRY(hipMemsetAsync(ctx->T, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->R, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->S, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
    return ASM_OK;
  }


  // eigenvalues of a symmetric tridiagonal matrix through the dense Jacobi solver
  double
  tridiag_max_eigenvalue(const std::vector<double> &diag, const std::vector<double> &off)
  {
    const int           k = diag.size();
    std::vector<double> M(k * k, 0.), K(k * k, 0.), lam, Q;
    for (int i = 0; i < k; ++i)
      {
        M[i * k + i] = 1.;
        K[i * k + i] = diag[i];
        if (i + 1 < k)
          K[i * k + i + 1] = K[(i + 1) * k + i] = off[i];
      }
    generalized_eigen(k, M, K, lam, Q);
    return lam.back();
  }
} // namespace

extern "C" {

void
asm_params_default(asm_params *p)
{
  std::memset(p, 0, sizeof(*p));
  p->degree = 4;
  for (int d = 0; d < 3; ++d)
    {
      p->n_cells[d]     = 8;
      p->boundary[d][0] = p->boundary[d][1] = ASM_BC_DIRICHLET;
      p->extent[d]                          = 1.0;
    }
  p->geometry         = ASM_GEOM_CARTESIAN;
  p->kershaw_eps[0]   = p->kershaw_eps[1] = 1.0;
  p->precond_type     = ASM_PRECOND_FDM;
  p->n_overlap        = 1;
  p->weighting_type   = ASM_W_POST; // ctor default, include/matrix_free.h:81-82
  p->element_centric  = 1;
  p->cheb_degree      = 3;
  p->smoothing_range  = 20.;
  p->eig_n_iterations = 40;
  p->ev_algorithm     = ASM_EV_AUTO;
  p->kernel_variant   = ASM_KERNEL_AUTO;
  p->device           = -1;
  p->stream           = nullptr;
  p->rank             = 0;
  p->world_size       = 1;
  p->polynomial_type  = ASM_POLY_FIRST_KIND;
  p->struct_size = (uint32_t)sizeof(asm_params);
}

int
asm_params_set(asm_params *p, const char *key, const char *value)
{
  if (p == nullptr || key == nullptr || value == nullptr)
    return ASM_ERR_INVALID_VALUE;
  const std::string k(key), v(value);
  const auto        fail = [&](const std::string &what) {
    g_create_error = what + " <" + v + "> is not known!"; // wording of precondition.templates.h:26, 130-133
    return ASM_ERR_INVALID_VALUE;
  };
  const auto to_int = [&](int32_t &out) {
    char *    end = nullptr;
    const long x  = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end != 0)
      return false;
    out = (int32_t)x;
    return true;
  };
  if (k == "fe degree")
    return to_int(p->degree) ? ASM_OK : fail("Degree");
  if (k == "n overlap")
    return to_int(p->n_overlap) ? ASM_OK : fail("Overlap");
  if (k == "degree")
    return to_int(p->cheb_degree) ? ASM_OK : fail("Chebyshev degree");
  if (k == "smoothing range")
    {
      char *end          = nullptr;
      p->smoothing_range = std::strtod(v.c_str(), &end);
      return (end != v.c_str() && *end == 0) ? ASM_OK : fail("Smoothing range");
    }
  if (k == "weighting type")
    {
      // get_weighting_type, precondition.templates.h:11-28
      static const char *names[] = {"none", "pre", "post", "ras", "symm"};
      for (int i = 0; i < 5; ++i)
        if (v == names[i])
          {
            p->weighting_type = i;
            return ASM_OK;
          }
      return fail("Weighting type");
    }
  if (k == "element centric")
    {
      if (v == "true" || v == "1")
        p->element_centric = 1;
      else if (v == "false" || v == "0")
        p->element_centric = 0;
      else
        return fail("Element centric");
      return ASM_OK;
    }
  if (k == "type")
    {
      if (v == "FDM")
        p->precond_type = ASM_PRECOND_FDM;
      else if (v == "Diagonal")
        p->precond_type = ASM_PRECOND_DIAGONAL;
      else
        return fail("Preconditioner");
      return ASM_OK;
    }
  if (k == "ev algorithm")
    {
      if (v == "lanczos")
        p->ev_algorithm = ASM_EV_LANCZOS;
      else if (v == "power iteration")
        p->ev_algorithm = ASM_EV_POWER_ITERATION;
      else
        return fail("Eigen-value algorithm");
      return ASM_OK;
    }
  if (k == "polynomial type")
    {
      if (v == "1st kind")
        p->polynomial_type = ASM_POLY_FIRST_KIND;
      else if (v == "4th kind")
        p->polynomial_type = ASM_POLY_FOURTH_KIND;
      else
        return fail("Polynomial type");
      return ASM_OK;
    }
  // Keys of the reference's preconditioner block that choose between implementations of the same mathematical
  // operation there (precondition.templates.h:195-211, 341, 388, 534): this library has one implementation of each, so
  // every value the reference accepts is taken and changes nothing - a driver can forward the reference's JSON block
  // verbatim.  Values the reference itself would reject are rejected.
  const auto is_bool = [&]() { return v == "true" || v == "false" || v == "1" || v == "0"; };
  if (k == "reuse partitioner") // one partitioner (box layout) serves the operator and the patches
    return is_bool() ? ASM_OK : fail("Reuse partitioner");
  if (k == "overlap pre post") // the ghost exchange overlaps the sweep whenever a block has interfaces (ASM_NO_OVERLAP)
    return is_bool() ? ASM_OK : fail("Overlap pre post");
  if (k == "weight sequence") // where the weights are multiplied in: matrix_free.h:84-94, 717-779; the same numbers
    return (v == "global" || v == "compressed" || v == "dg") ? ASM_OK : fail("Weight sequence");
  if (k == "optimize") // how much of the Chebyshev iteration is fused into the sweeps (0-3): always everything
    {
      int32_t o = 0;
      return (to_int(o) && o >= 0 && o <= 3) ? ASM_OK : fail("Optimize");
    }
  if (k == "sub mesh approximation")
    {
      // cells per direction of the sub-mesh the patch matrices are assembled from: the reference's default (dim = 3,
      // the full neighbourhood) is what setup.h implements
      int32_t n = 0;
      if (!to_int(n) || n < 1 || n > 3)
        return fail("Sub mesh approximation");
      if (n != 3)
        {
          g_create_error = "sub mesh approximation <" + v + "> is not implemented (only the reference's default, 3)";
          return ASM_ERR_NOT_SUPPORTED;
        }
      return ASM_OK;
    }
  g_create_error = "unknown option key <" + k + ">";
  return ASM_ERR_INVALID_VALUE;
}

const char *
asm_last_error(const asm_ctx *ctx)
{
  return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int
asm_describe_layout(const asm_params *p, asm_layout *out)
{
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  fill_layout(s, out);
  return ASM_OK;
}

int
asm_tiled_plan(const asm_params *p, int32_t n_cu, int32_t rows, int32_t plan[8], int32_t *order, int32_t max_order, int32_t *n_order)
{
  if (p == nullptr || plan == nullptr || n_order == nullptr)
    return ASM_ERR_INVALID_VALUE;
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  if (s.g.p != 4)
    {
      g_create_error = "asm_tiled_plan: the tiled sweeps exist for degree 4";
      return ASM_ERR_NOT_SUPPORTED;
    }
  int out[8];
  tiled_plan(s.g, n_cu, rows, out);
  for (int i = 0; i < 8; ++i)
    plan[i] = out[i];
  std::vector<int> ord;
  *n_order = tiled_order(s.g, n_cu, rows, ord);
  if (order != nullptr)
    for (int i = 0; i < std::min<int>(max_order, (int)ord.size()); ++i)
      order[i] = ord[i];
  return ASM_OK;
}

int
asm_exchange_plan(const asm_params *p, int add, asm_exchange_step steps[3], int32_t *n_steps)
{
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  *n_steps = build_exchange_plan(s, add != 0, steps);
  return ASM_OK;
}

int
asm_exchange_plan_direct(const asm_params *p, asm_direct_message msgs[26], int32_t *n_msgs)
{
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  *n_msgs = build_direct_plan(s, msgs);
  return ASM_OK;
}

int
asm_create(const asm_params *p, asm_ctx **out)
{
  *out = nullptr;
  std::unique_ptr<asm_ctx> holder(new asm_ctx);
  asm_ctx *                ctx = holder.get();
  const auto               fail = [&](const int code) {
    g_create_error = ctx->error;
    for (void *q : ctx->allocs)
      (void)hipFree(q);
    if (ctx->h_scalar)
      (void)hipHostFree(ctx->h_scalar);
    (void)hipGetLastError();
    return code;
  };
  if (!setup_layout(*p, ctx->s))
    {
      ctx->error = ctx->s.error;
      return fail(ASM_ERR_INVALID_VALUE);
    }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
    {
      ctx->error = "no HIP device available (the MI355X path has no CPU fallback)";
      return fail(ASM_ERR_NO_DEVICE);
    }
  if (p->device >= n_dev || (p->device < 0 && hipGetDevice(&ctx->device) != hipSuccess))
    {
      ctx->error = "invalid HIP device ordinal";
      return fail(ASM_ERR_DEVICE);
    }
  if (p->device >= 0)
    ctx->device = p->device;
  const DeviceScope device_scope(ctx);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess)
      {
        ctx->error = "hipGetDeviceProperties failed";
        return fail(ASM_ERR_DEVICE);
      }
    ctx->n_cu = prop.multiProcessorCount;
  }
  ctx->stream = (hipStream_t)p->stream;
  try
    {
      ctx->basis.reset(new Basis1D(p->degree));
      if (!setup_all(ctx->s, *ctx->basis))
        {
          ctx->error = ctx->s.error;
          return fail(ASM_ERR_INVALID_VALUE);
        }
    }
  catch (const std::exception &e)
    {
      ctx->error = e.what();
      return fail(ASM_ERR_INVALID_VALUE);
    }
  HostSetup &s = ctx->s;
  ctx->g       = s.g;
  int err;
#define UP(vec, ptr)                 \
  err = upload(ctx, vec, ptr);       \
  if (err)                           \
    return fail(err);
  // operator data
  DevOp &op    = ctx->op;
  op.cartesian = s.cartesian;
  for (int d = 0; d < 3; ++d)
    {
      UP(s.h[d], &op.h[d]);
    }
  UP(s.coeff, &op.coeff);
  UP(ctx->basis->N, &op.N);
  UP(ctx->basis->Dc, &op.Dc);
  UP(ctx->basis->xq, &op.xq);
  UP(ctx->basis->wq, &op.wq);
  if (ctx->basis->n1 == 5)
    ctx->basis->diagonalise();
  UP(ctx->basis->F_ref, &op.F_ref);
  UP(ctx->basis->lam_ref, &op.lam_ref);
  // low-order levels (the bottom of a multigrid hierarchy): one thread per cell (ASM_NO_SMALL keeps the generic sweeps)
  const bool small_sweeps = std::getenv("ASM_NO_SMALL") == nullptr;
  op.small                = small_sweeps && s.cartesian && s.g.p <= 2;
  for (int q = 0; q < 9; ++q)
    op.M_small[q] = op.K_small[q] = 0.;
  if (op.small)
    for (int q = 0; q < ctx->basis->n1 * ctx->basis->n1; ++q)
      {
        op.M_small[q] = ctx->basis->M_ref[q];
        op.K_small[q] = ctx->basis->K_ref[q];
      }
  // wave-pair sweeps: the tables of A (tiled Laplace contexts) and of P (tiled FDM contexts on tensor-product meshes)
  {
    const char *e        = std::getenv("ASM_PAIR");
    const bool  use_pair = e == nullptr || std::atoi(e) != 0;
    const auto  up_kron  = [&](const HostKron &hk, DevKron &dk) -> int {
      dk.valid = 0;
      if (!hk.valid)
        return ASM_OK;
      for (int d = 0; d < 3; ++d)
        {
          dk.n_cls[d] = hk.n_cls[d];
          UP(hk.cls[d], &dk.cls[d]);
        }
      UP(hk.sym, &dk.sym);
      UP(hk.mat, &dk.mat);
      UP(hk.diag, &dk.diag);
      dk.mask_constrained = hk.mask_constrained ? 1 : 0;
      dk.valid            = 1;
      return ASM_OK;
    };
    HostKron hk;
    if (use_pair && s.laplace_tiled && build_kron_laplace(s, *ctx->basis, hk))
      {
        err = up_kron(hk, ctx->kron_a);
        if (err)
          return fail(err);
      }
    if (use_pair && s.fdm_tiled && s.prm.precond_type == ASM_PRECOND_FDM && build_kron_fdm(s, hk))
      {
        err = up_kron(hk, ctx->kron_p);
        if (err)
          return fail(err);
      }
  }
  // FDM data
  DevFdm &f = ctx->f;
  std::memset(&f, 0, sizeof(f));
  f.small       = small_sweeps && s.m <= 3;
  f.m           = s.m;
  const bool vx = s.prm.element_centric == 0;
  f.origin_mul  = s.g.p;
  f.origin_add  = vx ? 1 : -(s.prm.n_overlap - 1);
  f.vertex      = vx;
  f.tensor_inst = s.tensor_inst;
  if (s.prm.precond_type == ASM_PRECOND_FDM)
    {
      for (int d = 0; d < 3; ++d)
        {
          UP(s.inst[d], &f.inst[d]);
          UP(s.w_pre[d], &f.w_pre[d]);
          UP(s.w_post[d], &f.w_post[d]);
        }
      UP(s.eig, &f.eig);
      UP(s.eig_sym, &f.eig_sym);

      f.n_inst   = s.n_inst;
      f.has_pre  = s.has_pre;
      f.has_post = s.has_post;
      if (s.tensor_inst && s.n_inst <= 8)
        {
          // reciprocal eigenvalue sums of every instance triple (a few hundred KB, cache resident)
          const int           m = s.m, E = m * m + m, n = s.n_inst;
          std::vector<double> inv((size_t)n * n * n * m * m * m);
          for (int iz = 0; iz < n; ++iz)
            for (int iy = 0; iy < n; ++iy)
              for (int ix = 0; ix < n; ++ix)
                for (int c = 0; c < m; ++c)
                  for (int b = 0; b < m; ++b)
                    for (int a = 0; a < m; ++a)
                      inv[(((size_t)(iz * n + iy) * n + ix) * m + c) * m * m + b * m + a] =
                        1.0 / (s.eig[(size_t)ix * E + m * m + a] + (s.eig[(size_t)iy * E + m * m + b] + s.eig[(size_t)iz * E + m * m + c]));
          UP(inv, &f.inv_table);
        }
      // 7^3 patches on a tensor-product mesh: the box of cells that share one parity-sorted instance per direction
      // (all cells but those next to a boundary) goes to the slice-per-lane kernel (kernels_slice7.hip)
      f.s7_on = 0;
      if (s.tensor_inst && s.m == 7 && !s.fdm_tiled && std::getenv("ASM_NO_SLICE7") == nullptr)
        {
          bool ok = true;
          for (int d = 0; d < 3 && ok; ++d)
            {
              // longest run of cells with one parity-sorted instance and the same rows of weight factors
              // (the kernel reads both as scalar operands)
              const auto same = [&](const int c0, const int c1) {
                if (s.inst[d][c0 + 1] != s.inst[d][c1 + 1])
                  return false;
                for (int i = 0; i < 7; ++i)
                  if (s.w_pre[d][(c0 + 1) * 7 + i] != s.w_pre[d][(c1 + 1) * 7 + i] || s.w_post[d][(c0 + 1) * 7 + i] != s.w_post[d][(c1 + 1) * 7 + i])
                    return false;
                return true;
              };
              int best_lo = 0, best_len = 0;
              for (int c = 0; c < s.g.nc[d];)
                {
                  const int id = s.inst[d][c + 1];
                  int       e  = c;
                  while (e < s.g.nc[d] && same(c, e))
                    ++e;
                  if (s.eig_sym[id] != 0 && e - c > best_len)
                    {
                      best_len = e - c;
                      best_lo  = c;
                    }
                  c = e;
                }
              ok            = best_len > 0;
              f.s7_lo[d]    = best_lo;
              f.s7_hi[d]    = best_lo + best_len;
              f.s7_inst[d]  = ok ? s.inst[d][best_lo + 1] : 0;
            }
          if (ok)
            {
              const int           m = 7, E = m * m + m;
              // in the order the slice kernel scales in: [x index a][z index c][y index b]
              std::vector<double> inv((size_t)m * m * m);
              for (int a = 0; a < m; ++a)
                for (int c = 0; c < m; ++c)
                  for (int b = 0; b < m; ++b)
                    inv[(a * m + c) * m + b] = 1.0 / (s.eig[(size_t)f.s7_inst[0] * E + m * m + a] +
                                                      (s.eig[(size_t)f.s7_inst[1] * E + m * m + b] + s.eig[(size_t)f.s7_inst[2] * E + m * m + c]));
              UP(inv, &f.s7_inv);
              for (int k = 0; k < 2; ++k)
                {
                  const std::vector<double> &wy = k == 0 ? s.w_pre[1] : s.w_post[1], &wz = k == 0 ? s.w_pre[2] : s.w_post[2];
                  std::vector<double>        wyz(49);
                  for (int kk = 0; kk < 7; ++kk)
                    for (int j = 0; j < 7; ++j)
                      wyz[kk * 7 + j] = wy[(f.s7_lo[1] + 1) * 7 + j] * wz[(f.s7_lo[2] + 1) * 7 + kk];
                  UP(wyz, &f.s7_wyz[k]);
                }
              std::vector<int> others;
              for (int cz = 0; cz < s.g.nc[2]; ++cz)
                for (int cy = 0; cy < s.g.nc[1]; ++cy)
                  for (int cx = 0; cx < s.g.nc[0]; ++cx)
                    if (!(cx >= f.s7_lo[0] && cx < f.s7_hi[0] && cy >= f.s7_lo[1] && cy < f.s7_hi[1] && cz >= f.s7_lo[2] && cz < f.s7_hi[2]))
                      others.push_back(cx + s.g.nc[0] * (cy + s.g.nc[1] * cz));
              f.s7_n_others = (int)others.size();
              f.s7_others   = nullptr;
              if (!others.empty())
                {
                  UP(others, &f.s7_others);
                }
              f.s7_on = 1;
            }
        }
      // 7^3 patches on a mesh without tensor-product structure (deformed cells): the box of cells all of whose patch DoFs
      // exist goes to the per-cell-matrix variant of the slice-per-lane kernel
      if (!s.tensor_inst && s.m == 7 && !s.fdm_tiled && s.prm.world_size == 1 && std::getenv("ASM_NO_SLICE7") == nullptr)
        {
          bool ok = true;
          for (int d = 0; d < 3; ++d)
            {
              int lo = 0, hi = s.g.nc[d] - 1;
              while (lo <= hi && f.origin_mul * lo + f.origin_add < s.g.ext_lo[d])
                ++lo;
              while (hi >= lo && f.origin_mul * hi + f.origin_add + 6 > s.g.ext_hi[d])
                --hi;
              f.s7_lo[d]   = lo;
              f.s7_hi[d]   = hi + 1;
              f.s7_inst[d] = 0;
              ok           = ok && hi >= lo;
            }
          if (ok)
            {
              std::vector<int> others;
              for (int cz = 0; cz < s.g.nc[2]; ++cz)
                for (int cy = 0; cy < s.g.nc[1]; ++cy)
                  for (int cx = 0; cx < s.g.nc[0]; ++cx)
                    if (!(cx >= f.s7_lo[0] && cx < f.s7_hi[0] && cy >= f.s7_lo[1] && cy < f.s7_hi[1] && cz >= f.s7_lo[2] && cz < f.s7_hi[2]))
                      others.push_back(cx + s.g.nc[0] * (cy + s.g.nc[1] * cz));
              f.s7_n_others = (int)others.size();
              f.s7_others   = nullptr;
              if (!others.empty())
                {
                  UP(others, &f.s7_others);
                }
              f.s7_inv = nullptr;
              f.s7_wyz[0] = f.s7_wyz[1] = nullptr;
              f.s7_on     = 2;
            }
        }
      // accumulate form of the 7^3 sweep (kernels.h: fdm_slice7_acc), one rank: which DoF planes belong to one column /
      // z chunk of the box alone
      f.s7_flag[0] = f.s7_flag[1] = f.s7_flag[2] = nullptr;
      f.s7_lz                                    = 0;
      if (f.s7_on && f.origin_mul == 4 && s.prm.world_size == 1 && !s.self_interface && s.prm.weighting_type != ASM_W_RAS &&
          std::getenv("ASM_S7_TMP") == nullptr) // (the predicate of multi_rank(): the accumulate form runs on one rank only)
        {
          const int n2 = f.s7_hi[2] - f.s7_lo[2];
          const int columns = ((f.s7_hi[0] - f.s7_lo[0] + S7_TILE_X - 1) / S7_TILE_X) * ((f.s7_hi[1] - f.s7_lo[1] + S7_TILE_Y - 1) / S7_TILE_Y);
          // about eight workgroups per compute unit (two are resident), chunks of at least six layers (a chunk shares
          // three of its 4 lz + 3 DoF planes with the next)
          const int want = (8 * std::max(ctx->n_cu, 1) + columns - 1) / columns;
          int       lz   = std::max(6, (n2 + want - 1) / want);
          if (const char *e = std::getenv("ASM_S7_LZ"))
            lz = std::max(1, std::atoi(e));
          f.s7_lz = std::min(lz, n2);
          for (int d = 0; d < 3; ++d)
            {
              const int                  n_store = s.g.wrap[d] ? s.g.nc[d] * s.g.p : s.g.ext_hi[d] - s.g.ext_lo[d] + 1;
              const int                  T = d == 0 ? S7_TILE_X : (d == 1 ? S7_TILE_Y : f.s7_lz), none = -(1 << 30);
              std::vector<int>           owner(n_store, none), coord(n_store, 0);
              std::vector<unsigned char> multi(n_store, 0), flag(n_store, 1);
              for (int c = 0; c < s.g.nc[d]; ++c)
                {
                  const int unit = (c >= f.s7_lo[d] && c < f.s7_hi[d]) ? (c - f.s7_lo[d]) / T : -1 - c;
                  for (int pos = 0; pos < 7; ++pos)
                    {
                      const int I = f.origin_mul * c + f.origin_add + pos;
                      if (!s.g.exists(d, I))
                        continue;
                      const int si = s.g.store_index(d, I);
                      if (owner[si] == none)
                        {
                          owner[si] = unit;
                          coord[si] = I;
                        }
                      else if (owner[si] != unit || coord[si] != I) // (a periodic direction meets itself around the block)
                        multi[si] = 1;
                    }
                }
              for (int si = 0; si < n_store; ++si)
                flag[si] = !(owner[si] >= 0 && !multi[si]);
              if (!s.g.wrap[d])
                {
                  if (s.g.dir_lo[d])
                    flag[0 - s.g.ext_lo[d]] = 1;
                  if (s.g.dir_hi[d])
                    flag[s.g.nc[d] * s.g.p - s.g.ext_lo[d]] = 1;
                }
              UP(flag, &f.s7_flag[d]);
            }
        }
    }
#undef UP
  if (alloc_vec(ctx, &ctx->R) || alloc_vec(ctx, &ctx->T) || alloc_vec(ctx, &ctx->S))
    return fail(ASM_ERR_DEVICE);
  ctx->allocs.push_back(ctx->R);
  ctx->allocs.push_back(ctx->T);
  ctx->allocs.push_back(ctx->S);
  // (restricted additive Schwarz keeps the atomics: only a patch's own DoFs have a non-zero weight,
  // so a sixth of the entries is added and the atomic form is the faster one - measured)
  ctx->ras_direct = std::getenv("ASM_RAS_ATOMICS") == nullptr;
  if (s.prm.precond_type == ASM_PRECOND_FDM && !s.fdm_tiled && s.prm.weighting_type != ASM_W_RAS && std::getenv("ASM_FDM_ATOMICS") == nullptr &&
      ctx->f.s7_flag[0] == nullptr)
    {
      // patch results of the generic FDM sweep; without it (allocation failure) the sweep adds with atomics
      const size_t bytes = (size_t)s.g.nc[0] * s.g.nc[1] * s.g.nc[2] * s.m * s.m * s.m * sizeof(double);
      if (hipMalloc((void **)&ctx->patch_tmp, bytes) == hipSuccess)
        {
          ctx->allocs.push_back(ctx->patch_tmp);
          ctx->data_bytes += bytes;
        }
      else
        {
          (void)hipGetLastError();
          ctx->patch_tmp = nullptr;
        }
    }
  if (hipMalloc((void **)&ctx->d_scalar, 64) != hipSuccess || hipHostMalloc((void **)&ctx->h_scalar, 64) != hipSuccess)
    {
      ctx->error = "allocation of reduction scalars failed";
      return fail(ASM_ERR_DEVICE);
    }
  ctx->allocs.push_back(ctx->d_scalar);
  if (multi_rank(ctx))
    {
      // largest slab: widest halo times the largest stored face
      const long long a = s.store_size[0], b = s.store_size[1], c = s.store_size[2];
      const long long face = std::max(a * b, std::max(a * c, b * c));
      ctx->hbuf_size       = face * (2 * s.g.p + 2);
      for (int i = 0; i < 4; ++i)
        {
          if (hipMalloc((void **)&ctx->hbuf[i], ctx->hbuf_size * sizeof(double)) != hipSuccess)
            {
              ctx->error = "halo buffer allocation failed";
              return fail(ASM_ERR_DEVICE);
            }
          ctx->allocs.push_back(ctx->hbuf[i]);
        }
      // single-round plan of the ghost update
      ctx->n_dmsg  = build_direct_plan(s, ctx->dmsg);
      ctx->dsend.n = ctx->drecv.n = ctx->n_dmsg;
      long long total[2]          = {0, 0};
      for (int i = 0; i < ctx->n_dmsg; ++i)
        {
          const asm_direct_message &m = ctx->dmsg[i];
          for (int d = 0; d < 3; ++d)
            {
              ctx->dsend.lo[i][d]   = m.send_lo[d];
              ctx->dsend.size[i][d] = m.send_size[d];
              ctx->drecv.lo[i][d]   = m.recv_lo[d];
              ctx->drecv.size[i][d] = m.recv_size[d];
            }
          ctx->dsend.offset[i] = m.send_offset;
          ctx->drecv.offset[i] = m.recv_offset;
          total[0] += m.send_count;
          total[1] += m.recv_count;
        }
      for (int i = 0; i < 2; ++i)
        {
          if (hipMalloc((void **)&ctx->dbuf[i], std::max(1ll, total[i]) * sizeof(double)) != hipSuccess)
            {
              ctx->error = "halo buffer allocation failed";
              return fail(ASM_ERR_DEVICE);
            }
          ctx->allocs.push_back(ctx->dbuf[i]);
        }
    }
  if (s.prm.precond_type == ASM_PRECOND_DIAGONAL)
    {
      if (alloc_vec(ctx, &ctx->inv_diag))
        return fail(ASM_ERR_DEVICE);
      ctx->allocs.push_back(ctx->inv_diag);
      if (!multi_rank(ctx)) // with several ranks the diagonal needs the communicator: done in asm_comm_init
        {
          err = asm_op_inverse_diagonal(ctx, ctx->inv_diag);
          if (err)
            return fail(err);
        }
    }
  if ((err = tune_tiled_laplace(ctx)))
    return fail(err);
  // single-phase tiled sweeps: the workgroups at Dirichlet faces first (kernels.h: tiled_order); ASM_NO_TILED_ORDER keeps
  // the launch positions (experiments)
  // (tune_tiled_laplace has set the tables of a context that measures its candidates; the others:)
  if (ctx->kron_a.valid || ctx->kron_p.valid)
    if ((err = upload_pair_orders(ctx, 0)))
      return fail(err);
  if ((ctx->s.laplace_tiled || ctx->s.fdm_tiled) && std::getenv("ASM_NO_TILED_ORDER") == nullptr && ctx->order_p == nullptr)
    {
      std::vector<int> order;
      tiled_order(ctx->g, ctx->n_cu, 0, order);
      if ((err = upload(ctx, order, &ctx->order_p)))
        return fail(err);
      ctx->order_a = ctx->order_p;
      if (ctx->s.laplace_tiled && ctx->laplace_rows != 0)
        {
          tiled_order(ctx->g, ctx->n_cu, ctx->laplace_rows, order);
          if ((err = upload(ctx, order, &ctx->order_a)))
            return fail(err);
        }
    }
  if ((err = tune_pair(ctx)))
    return fail(err);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess)
    {
      ctx->error = "device error during setup";
      return fail(ASM_ERR_DEVICE);
    }
  *out = holder.release();
  return ASM_OK;
}

void
asm_destroy(asm_ctx *ctx)
{
  if (!ctx)
    return;
  const DeviceScope device_scope(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm_stream)
    {
      (void)hipStreamSynchronize(ctx->comm_stream);
      (void)hipEventDestroy(ctx->ev_boundary);
      (void)hipEventDestroy(ctx->ev_exchanged);
      (void)hipStreamDestroy(ctx->comm_stream);
      if (ctx->bnd_stream)
        {
          (void)hipStreamSynchronize(ctx->bnd_stream);
          (void)hipEventDestroy(ctx->ev_start);
          (void)hipStreamDestroy(ctx->bnd_stream);
        }
    }
  if (ctx->comm && !ctx->comm_shared)
    ncclCommDestroy(ctx->comm);
  for (int k = 0; k < 2; ++k)
    for (auto &e : ctx->prof_events[k])
      {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
      }
  for (void *q : ctx->allocs)
    (void)hipFree(q);
  for (double *q : ctx->work_pool)
    (void)hipFree(q);
  if (ctx->h_scalar)
    (void)hipHostFree(ctx->h_scalar);
  delete ctx;
}

int
asm_comm_unique_id(char id[128])
{
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId uid;
  if (ncclGetUniqueId(&uid) != ncclSuccess)
    return ASM_ERR_DEVICE;
  std::memcpy(id, &uid, 128);
  return ASM_OK;
}

static int
setup_overlap(asm_ctx *ctx)
{
  if (ctx->s.kernel == ASM_KERNEL_TILED && ctx->n_dmsg > 0 && std::getenv("ASM_NO_OVERLAP") == nullptr && !ctx->overlap)
    {
      HIP_TRY(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_boundary, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_exchanged, hipEventDisableTiming));
      if (std::getenv("ASM_CONCURRENT_PHASES") != nullptr) // experiments: measured slower (cross-queue waits, profiles/r02_multirank_rehearsal.txt)
        {
          int lo = 0, hi = 0; // (numerically lower = higher priority)
          HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
          HIP_TRY(hipStreamCreateWithPriority(&ctx->bnd_stream, hipStreamNonBlocking, hi));
          HIP_TRY(hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming));
        }
      ctx->overlap = true;
    }
  return ASM_OK;
}

// testing aid: ctxs[r] = rank r of n; afterwards every rank must be driven by its own host thread
int
asm_debug_loopback_init(asm_ctx **ctxs, int n)
{
  auto lg = std::make_shared<LoopbackGroup>();
  lg->ranks.assign(ctxs, ctxs + n);
  lg->scalars.assign(n, 0.);
  lg->buffers.assign(n, nullptr);
  for (int r = 0; r < n; ++r)
    {
      asm_ctx *ctx = ctxs[r];
      if (ctx->s.prm.rank != r || ctx->s.prm.world_size != n)
        {
          ctx->error = "asm_debug_loopback_init: ctxs[r] must be rank r of n";
          return ASM_ERR_INVALID_VALUE;
        }
      ctx->loop     = lg;
      const int err = setup_overlap(ctx);
      if (err)
        return err;
    }
  return ASM_OK;
}

int
asm_comm_init(asm_ctx *ctx, const char id[128])
{
  const DeviceScope device_scope(ctx);
  if (ctx == nullptr || id == nullptr)
    {
      g_create_error = "asm_comm_init: context and unique id must be given";
      return ASM_ERR_INVALID_VALUE;
    }
  if (!multi_rank(ctx))
    {
      ctx->error = "asm_comm_init: the context was created for a single rank (world_size = 1)";
      return ASM_ERR_INVALID_VALUE;
    }
  if (ctx->comm != nullptr || ctx->loop)
    {
      ctx->error = "asm_comm_init: the context already has a communicator";
      return ASM_ERR_INVALID_VALUE;
    }
  ncclUniqueId uid;
  std::memcpy(&uid, id, 128);
  NCCL_TRY(ncclCommInitRank(&ctx->comm, ctx->s.prm.world_size, uid, ctx->s.prm.rank));
  const int err = setup_overlap(ctx);
  if (err)
    return err;
  if (ctx->s.prm.precond_type == ASM_PRECOND_DIAGONAL)
    return asm_op_inverse_diagonal(ctx, ctx->inv_diag);
  return ASM_OK;
}

int
asm_vector_layout(const asm_ctx *ctx, asm_layout *out)
{
  fill_layout(ctx->s, out);
  return ASM_OK;
}

int
asm_vector_alloc(asm_ctx *ctx, double **v)
{
  const DeviceScope device_scope(ctx);
  return alloc_vec(ctx, v);
}

int
asm_vector_free(asm_ctx *ctx, double *v)
{
  const DeviceScope device_scope(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream)
    HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
  if (v == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  HIP_TRY(hipFree(v));
  return ASM_OK;
}

// ---- internal entry points of the multi-rank multigrid hierarchy (multigrid.hip; not part of the public interface) ----
// update_ghost_values (add = 0) / compress(add) + zero_out_ghost_values (add = 1) of a level vector
int
asm_internal_exchange(asm_ctx *ctx, double *v, int add)
{
  const DeviceScope device_scope(ctx);
  return exchange(ctx, v, add != 0);
}

// hipGraph capture of the V-cycle's coarse legs: the level contexts issue their work on `stream` while it is being
// captured; returns the previous stream
void *
asm_internal_swap_stream(asm_ctx *ctx, void *stream)
{
  void *old   = (void *)ctx->stream;
  ctx->stream = (hipStream_t)stream;
  return old;
}

// what a captured Chebyshev sweep of this context has baked into its kernel arguments: a graph is replayed only while
// these values stand
void
asm_internal_cheb_state(const asm_ctx *ctx, double out[4])
{
  out[0] = ctx->ev_set ? 1. : 0.;
  out[1] = ctx->theta;
  out[2] = ctx->delta;
  out[3] = ctx->s.prm.cheb_degree + 16. * ctx->s.prm.polynomial_type;
}

// zero_out_ghost_values
int
asm_internal_zero_ghosts(asm_ctx *ctx, double *v)
{
  const DeviceScope device_scope(ctx);
  if (multi_rank(ctx) && ctx->n_dmsg > 0)
    BOX_TRY(vec_boxes(ctx->g, 3, v, nullptr, ctx->drecv, ctx->stream));
  return launch_status(ctx, "zero ghosts");
}

// sum of the n doubles at v over the ranks of ctx's communicator, in place (a vector of a level that every rank holds whole)
int
asm_internal_allreduce(asm_ctx *ctx, double *v, long long n)
{
  const DeviceScope device_scope(ctx);
  if (!multi_rank(ctx) || n <= 0)
    return ASM_OK;
  if (ctx->loop)
    {
      LoopbackGroup &lg = *ctx->loop;
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      lg.buffers[ctx->s.prm.rank] = v;
      lg.barrier();
      std::vector<double> sum((size_t)n, 0.), part((size_t)n);
      for (const double *q : lg.buffers)
        {
          HIP_TRY(hipMemcpy(part.data(), q, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
          for (long long i = 0; i < n; ++i)
            sum[(size_t)i] += part[(size_t)i];
        }
      lg.barrier(); // every rank has read every buffer
      HIP_TRY(hipMemcpy(v, sum.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      return ASM_OK;
    }
  if (ctx->comm == nullptr)
    {
      ctx->error = "world_size > 1 but the hierarchy has no communicator (asm_mg_comm_init)";
      return ASM_ERR_INVALID_VALUE;
    }
  if (ctx->exchange_pending)
    {
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_exchanged, 0));
      ctx->exchange_pending = false;
    }
  NCCL_TRY(ncclAllReduce(v, v, (size_t)n, ncclDouble, ncclSum, ctx->comm, ctx->stream));
  return ASM_OK;
}

// the level contexts of a hierarchy use ONE communicator (the finest level's): operations of one communicator must not
// be in flight on two streams, so none of them overlaps its exchange with a sweep (asm_comm_init of `src` may have
// switched that on)
int
asm_internal_comm_share(asm_ctx *dst, asm_ctx *src)
{
  if (!multi_rank(dst))
    return ASM_OK;
  const DeviceScope device_scope(dst);
  src->overlap = false;
  if (dst != src)
    {
      if (dst->comm != nullptr || dst->loop)
        {
          dst->error = "the level context already has a communicator";
          return ASM_ERR_INVALID_VALUE;
        }
      dst->comm        = src->comm;
      dst->comm_shared = true;
      dst->overlap     = false;
    }
  if (dst->s.prm.precond_type == ASM_PRECOND_DIAGONAL)
    return asm_op_inverse_diagonal(dst, dst->inv_diag);
  return ASM_OK;
}

// Work vectors of the Krylov solvers (multigrid.hip): taken from / returned to a per-context pool, as deal.II's
// GrowingVectorMemory does for SolverGMRES / SolverCG, so that a repeated solve does not allocate.
int
asm_work_vector_acquire(asm_ctx *ctx, double **v)
{
  const DeviceScope device_scope(ctx);
  if (!ctx->work_pool.empty())
    {
      *v = ctx->work_pool.back();
      ctx->work_pool.pop_back();
      HIP_TRY(hipMemsetAsync(*v, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
      return ASM_OK;
    }
  return alloc_vec(ctx, v);
}

void
asm_work_vector_release(asm_ctx *ctx, double *v)
{
  if (v == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  ctx->work_pool.push_back(v);
}

int
asm_vector_import(asm_ctx *ctx, double *dev, const double *host_owned)
{
  const DeviceScope device_scope(ctx);
  const DevGrid &g = ctx->g;
  const int      lo[3] = {0, 0, 0}, size[3] = {g.own_hi[0] + 1, g.own_hi[1] + 1, g.own_hi[2] + 1};
  double *       buf   = nullptr;
  if (ctx->comm_stream)
    HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
  ctx->exchange_pending = false;
  if (dev == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, ctx->s.n_owned * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(buf, host_owned, ctx->s.n_owned * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(dev, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
  BOX_TRY(vec_box(g, 1, dev, buf, lo, size, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipFree(buf));
  return ASM_OK;
}

int
asm_vector_export(asm_ctx *ctx, double *host_owned, const double *dev)
{
  const DeviceScope device_scope(ctx);
  const DevGrid &g = ctx->g;
  const int      lo[3] = {0, 0, 0}, size[3] = {g.own_hi[0] + 1, g.own_hi[1] + 1, g.own_hi[2] + 1};
  double *       buf   = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, ctx->s.n_owned * sizeof(double)));
  BOX_TRY(vec_box(g, 0, const_cast<double *>(dev), buf, lo, size, ctx->stream));
  HIP_TRY(hipMemcpyAsync(host_owned, buf, ctx->s.n_owned * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipFree(buf));
  return ASM_OK;
}

int
asm_update_ghost_values(asm_ctx *ctx, double *v)
{
  const DeviceScope device_scope(ctx);
  return exchange(ctx, v, false);
}

int
asm_compress_add(asm_ctx *ctx, double *v)
{
  const DeviceScope device_scope(ctx);
  return exchange(ctx, v, true);
}

int
asm_vec_dot(asm_ctx *ctx, const double *x, const double *y, double *result)
{
  const DeviceScope device_scope(ctx);
  return dot(ctx, x, y, result);
}

int
asm_vec_multi_dot(asm_ctx *ctx, const double *w, const double *const *v, int32_t n, double *result)
{
  const DeviceScope device_scope(ctx);
  if (n < 0 || (n > 0 && (v == nullptr || result == nullptr)))
    {
      ctx->error = "asm_vec_multi_dot: bad arguments";
      return ASM_ERR_INVALID_VALUE;
    }
  const int err = multi_dot(ctx, w, v, n, result);
  return err ? err : launch_status(ctx, "asm_vec_multi_dot");
}

int
asm_vec_multi_axpy_norm(asm_ctx *ctx, double *w, const double *const *v, const double *a, int32_t n, double *norm2)
{
  const DeviceScope device_scope(ctx);
  if (n < 0 || norm2 == nullptr || (n > 0 && (v == nullptr || a == nullptr)))
    {
      ctx->error = "asm_vec_multi_axpy_norm: bad arguments";
      return ASM_ERR_INVALID_VALUE;
    }
  const int err = multi_axpy_norm(ctx, w, v, a, n, norm2);
  return err ? err : launch_status(ctx, "asm_vec_multi_axpy_norm");
}

// elementwise updates also rewrite the ghost planes: a vector marked as exchanged stays consistent only
// if every operand was, which the library does not track - drop the marker of the written vector
int
asm_vec_axpby(asm_ctx *ctx, double *y, double a, const double *x, double b)
{
  const DeviceScope device_scope(ctx);
  if (y == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  vec_axpby(y, a, x, b, ctx->s.n_alloc, ctx->stream);
  return launch_status(ctx, "asm_vec_axpby");
}

int
asm_vec_scale(asm_ctx *ctx, double *y, double a, const double *x)
{
  const DeviceScope device_scope(ctx);
  if (y == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  vec_scale(y, x, a, ctx->s.n_alloc, ctx->stream);
  return launch_status(ctx, "asm_vec_scale");
}

int
asm_vec_copy(asm_ctx *ctx, double *y, const double *x)
{
  const DeviceScope device_scope(ctx);
  if (y == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  if (y != x)
    HIP_TRY(hipMemcpyAsync(y, x, ctx->s.n_alloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return ASM_OK;
}

int
asm_vec_zero(asm_ctx *ctx, double *y)
{
  const DeviceScope device_scope(ctx);
  if (y == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  HIP_TRY(hipMemsetAsync(y, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
  return ASM_OK;
}

int
asm_vec_constraints_set_zero(asm_ctx *ctx, double *y)
{
  const DeviceScope device_scope(ctx);
  vec_constrained(ctx->g, 0, y, nullptr, nullptr, ctx->stream);
  return launch_status(ctx, "asm_vec_constraints_set_zero");
}

int
asm_op_vmult(asm_ctx *ctx, double *dst, const double *src)
{
  const DeviceScope device_scope(ctx);
  if (!ctx->keep_ghosts)
    ctx->exchanged_vec = nullptr;
  return op_apply(ctx, OP_PLAIN, dst, const_cast<double *>(src), nullptr);
}

int
asm_op_inverse_diagonal(asm_ctx *ctx, double *out)
{
  const DeviceScope device_scope(ctx);
  if (out == ctx->exchanged_vec)
    ctx->exchanged_vec = nullptr;
  HIP_TRY(hipMemsetAsync(out, 0, ctx->s.n_alloc * sizeof(double), ctx->stream));
  if (!laplace_generic(ctx->g, ctx->op, nullptr, out, true, ctx->stream))
    {
      ctx->error = "generic Laplace kernel not available for this degree";
      return ASM_ERR_NOT_SUPPORTED;
    }
  const int err = exchange(ctx, out, true);
  if (err)
    return err;
  vec_invert_diagonal(out, ctx->s.n_alloc, ctx->stream);
  return launch_status(ctx, "compute_inverse_diagonal");
}

int
asm_precond_vmult(asm_ctx *ctx, double *dst, const double *src)
{
  const DeviceScope device_scope(ctx);
  if (!ctx->keep_ghosts)
    ctx->exchanged_vec = nullptr;
  return pc_apply(ctx, PC_PLAIN, dst, const_cast<double *>(src), nullptr, nullptr, 0., 1.);
}

int
asm_op_vmult_fused(asm_ctx *ctx, int mode, double *dst, const double *src, const double *b)
{
  const DeviceScope device_scope(ctx);
  if (mode != ASM_FUSED_NONE && mode != ASM_FUSED_RESIDUAL)
    {
      ctx->error = "asm_op_vmult_fused: mode must be ASM_FUSED_NONE or ASM_FUSED_RESIDUAL";
      return ASM_ERR_INVALID_VALUE;
    }
  if (dst == src || (mode == ASM_FUSED_RESIDUAL && b == nullptr))
    {
      ctx->error = "asm_op_vmult_fused: dst must not alias src; ASM_FUSED_RESIDUAL needs b";
      return ASM_ERR_INVALID_VALUE;
    }
  if (!ctx->keep_ghosts)
    ctx->exchanged_vec = nullptr;
  return op_apply(ctx, mode == ASM_FUSED_RESIDUAL ? OP_RESIDUAL : OP_PLAIN, dst, const_cast<double *>(src), b);
}

int
asm_precond_vmult_fused(asm_ctx *ctx, int mode, double *dst, const double *src, const double *x, const double *x_old, double f1,
                        double f2)
{
  const DeviceScope device_scope(ctx);
  int               m;
  switch (mode)
    {
      case ASM_FUSED_NONE: m = PC_PLAIN; break;
      case ASM_FUSED_CHEB_FIRST: m = PC_FIRST; break;
      case ASM_FUSED_RELAX: m = PC_STEP1; break;
      case ASM_FUSED_CHEB_NEXT: m = PC_UPDATE; break;
      default: ctx->error = "asm_precond_vmult_fused: unknown mode"; return ASM_ERR_INVALID_VALUE;
    }
  if (dst == src || ((m == PC_STEP1 || m == PC_UPDATE) && x == nullptr) || (m == PC_UPDATE && x_old == nullptr))
    {
      ctx->error = "asm_precond_vmult_fused: dst must not alias src; the update's operand vectors must be given";
      return ASM_ERR_INVALID_VALUE;
    }
  if (m == PC_UPDATE && dst == x)
    {
      ctx->error = "asm_precond_vmult_fused: ASM_FUSED_CHEB_NEXT may alias dst with x_old, not with x";
      return ASM_ERR_INVALID_VALUE;
    }
  if (!ctx->keep_ghosts)
    ctx->exchanged_vec = nullptr;
  return pc_apply(ctx, m, dst, const_cast<double *>(src), x, x_old, f1, m == PC_PLAIN ? 1. : f2);
}

int
asm_is_symmetric(const asm_ctx *ctx, int32_t *symmetric)
{
  const asm_params &prm = ctx->s.prm;
  *symmetric = (prm.precond_type == ASM_PRECOND_DIAGONAL || prm.weighting_type == ASM_W_NONE || prm.weighting_type == ASM_W_SYMM) ? 1 : 0;
  return ASM_OK;
}

int
asm_cheb_set_eigenvalue(asm_ctx *ctx, double lambda_max)
{
  if (!(lambda_max > 0.))
    {
      ctx->error = "eigenvalue estimate must be positive";
      return ASM_ERR_INVALID_VALUE;
    }
  // deal.II: max_eigenvalue = 1.2 * estimate; min = max / smoothing_range
  ctx->lambda_max     = lambda_max;
  const double max_ev = 1.2 * lambda_max;
  const double min_ev = max_ev / ctx->s.prm.smoothing_range;
  ctx->theta          = 0.5 * (max_ev + min_ev);
  // the 4th-kind polynomial only needs the upper bound: delta carries it
  ctx->delta          = ctx->s.prm.polynomial_type == ASM_POLY_FOURTH_KIND ? max_ev : 0.5 * (max_ev - min_ev);
  ctx->ev_set         = true;
  return ASM_OK;
}

int
asm_cheb_estimate_eigenvalues(asm_ctx *ctx, double *lambda_max)
{
  const DeviceScope device_scope(ctx);
  const asm_params &prm = ctx->s.prm;
  const long long   n   = ctx->s.n_alloc;
  int               alg = prm.ev_algorithm;
  ctx->exchanged_vec    = nullptr; // the vectors of the estimate are also written by plain vector kernels
  if (alg == ASM_EV_AUTO)
    {
      // SymmetryType, include/matrix_free.h:896-904 and precondition.templates.h:113-114
      const bool symmetric = prm.precond_type == ASM_PRECOND_DIAGONAL || prm.weighting_type == ASM_W_NONE ||
                             prm.weighting_type == ASM_W_SYMM;
      alg = symmetric ? ASM_EV_LANCZOS : ASM_EV_POWER_ITERATION;
    }
  double *v = nullptr, *w = nullptr, *z = nullptr, *q = nullptr;
  const auto cleanup = [&](const int code) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(v);
    (void)hipFree(w);
    (void)hipFree(z);
    (void)hipFree(q);
    return code;
  };
  if (alloc_vec(ctx, &v) || alloc_vec(ctx, &w) || alloc_vec(ctx, &z) || alloc_vec(ctx, &q))
    return cleanup(ASM_ERR_DEVICE);
  // set_initial_guess [deal.II]: (global index % 11) - mean, constrained entries zeroed
  long long gbegin[3], gsize[3];
  for (int d = 0; d < 3; ++d)
    {
      gbegin[d] = (long long)ctx->s.cell_begin[d] * prm.degree;
      gsize[d]  = (long long)ctx->s.ncg[d] * prm.degree + (prm.boundary[d][0] == ASM_BC_PERIODIC ? 0 : 1);
    }
  const long long ng  = ctx->s.n_global;
  const long long sum = (ng / 11) * 55 + ((ng % 11) * ((ng % 11) - 1)) / 2;
  vec_initial_guess(ctx->g, v, gbegin, gsize, ctx->stream);
  vec_add_scalar_owned(ctx->g, v, -(double)sum / (double)ng, ctx->stream);
  vec_constrained(ctx->g, 0, v, nullptr, nullptr, ctx->stream);
  int    err;
  double est = 0.;
  if (alg == ASM_EV_POWER_ITERATION)
    {
      double nrm2;
      if ((err = dot(ctx, v, v, &nrm2)))
        return cleanup(err);
      vec_scale(v, v, 1. / std::sqrt(nrm2), n, ctx->stream);
      for (int it = 0; it < prm.eig_n_iterations; ++it)
        {
          if ((err = op_apply(ctx, OP_PLAIN, w, v, nullptr)))
            return cleanup(err);
          if ((err = pc_apply(ctx, PC_PLAIN, z, w, nullptr, nullptr, 0., 1.)))
            return cleanup(err);
          if ((err = dot(ctx, v, z, &est)))
            return cleanup(err);
          if ((err = dot(ctx, z, z, &nrm2)))
            return cleanup(err);
          vec_scale(v, z, 1. / std::sqrt(nrm2), n, ctx->stream);
        }
      est = std::fabs(est);
    }
  else
    {
      // preconditioned CG started from x = 0, rhs = v; Lanczos matrix from the
      // CG coefficients (deal.II SolverCG eigenvalue slot)
      std::vector<double> alphas, betas;
      double *            r = v, *pv = q;
      double              rz, r0, tmp;
      if ((err = pc_apply(ctx, PC_PLAIN, z, r, nullptr, nullptr, 0., 1.)))
        return cleanup(err);
      if (hipMemcpyAsync(pv, z, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
        {
          ctx->error = "hipMemcpyAsync failed in the eigenvalue estimate";
          return cleanup(ASM_ERR_DEVICE);
        }
      if ((err = dot(ctx, r, z, &rz)) || (err = dot(ctx, r, r, &r0)))
        return cleanup(err);
      r0 = std::sqrt(r0);
      for (int it = 0; it < prm.eig_n_iterations; ++it)
        {
          if ((err = op_apply(ctx, OP_PLAIN, w, pv, nullptr)))
            return cleanup(err);
          if ((err = dot(ctx, pv, w, &tmp)))
            return cleanup(err);
          const double alpha = rz / tmp;
          alphas.push_back(alpha);
          vec_axpby(r, -alpha, w, 1., n, ctx->stream);
          if ((err = dot(ctx, r, r, &tmp)))
            return cleanup(err);
          if (std::sqrt(tmp) < 1e-7 * r0)
            break;
          if ((err = pc_apply(ctx, PC_PLAIN, z, r, nullptr, nullptr, 0., 1.)))
            return cleanup(err);
          double rz_new;
          if ((err = dot(ctx, r, z, &rz_new)))
            return cleanup(err);
          const double beta = rz_new / rz;
          betas.push_back(beta);
          rz = rz_new;
          vec_axpby(pv, 1., z, beta, n, ctx->stream);
        }
      const int           k = alphas.size();
      std::vector<double> dg(k), off(k > 0 ? k - 1 : 0);
      for (int i = 0; i < k; ++i)
        {
          dg[i] = 1. / alphas[i] + (i > 0 ? betas[i - 1] / alphas[i - 1] : 0.);
          if (i + 1 < k)
            off[i] = std::sqrt(betas[i]) / alphas[i];
        }
      est = k > 0 ? tridiag_max_eigenvalue(dg, off) : 1.;
    }
  *lambda_max = est;
  return cleanup(asm_cheb_set_eigenvalue(ctx, est));
}

int
asm_cheb_vmult(asm_ctx *ctx, double *dst, const double *src)
{
  const DeviceScope device_scope(ctx);
  return cheb_run(ctx, dst, src, true);
}

int
asm_cheb_step(asm_ctx *ctx, double *dst, const double *src)
{
  const DeviceScope device_scope(ctx);
  return cheb_run(ctx, dst, src, false);
}

int
asm_n_fdm_instances(const asm_ctx *ctx, int64_t *n)
{
  *n = ctx->s.n_inst;
  return ASM_OK;
}

int
asm_memory_consumption(const asm_ctx *ctx, int64_t *bytes)
{
  *bytes = ctx->data_bytes;
  return ASM_OK;
}

int
asm_cheb_parameters(const asm_ctx *ctx, double out[3])
{
  out[0] = ctx->theta;
  out[1] = ctx->delta;
  out[2] = ctx->lambda_max;
  return ASM_OK;
}

int
asm_kernel_variant_in_use(const asm_ctx *ctx, int32_t *variant)
{
  *variant = ctx->s.kernel;
  return ASM_OK;
}

int
asm_synchronize(asm_ctx *ctx)
{
  const DeviceScope device_scope(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream)
    HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
  return ASM_OK;
}

int
asm_set_keep_ghosts(asm_ctx *ctx, int on)
{
  ctx->keep_ghosts = on != 0;
  if (!ctx->keep_ghosts)
    ctx->exchanged_vec = nullptr;
  return ASM_OK;
}

int
asm_profile_enable(asm_ctx *ctx, int on)
{
  ctx->profile      = on != 0;
  ctx->prof_used[0] = ctx->prof_used[1] = 0;
  return ASM_OK;
}

int
asm_profile_get(asm_ctx *ctx, int kernel, int64_t *launches, double *total_ms)
{
  const DeviceScope device_scope(ctx);
  if (kernel < 0 || kernel > 1)
    {
      ctx->error = "asm_profile_get: kernel must be 0 (Laplace sweep) or 1 (FDM sweep)";
      return ASM_ERR_INVALID_VALUE;
    }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  double sum = 0.;
  for (size_t i = 0; i < ctx->prof_used[kernel]; ++i)
    {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_events[kernel][i].first, ctx->prof_events[kernel][i].second));
      sum += ms;
    }
  *launches = (int64_t)ctx->prof_used[kernel];
  *total_ms = sum;
  return ASM_OK;
}

// which sweeps of a tiled context take the wave-pair kernels (kernels_pair.hip; measured at creation against the tiled
// kernels, or fixed by ASM_PAIR): *laplace / *fdm = 1 or 0
int
asm_pair_sweeps_in_use(const asm_ctx *ctx, int32_t *laplace, int32_t *fdm)
{
  if (!ctx || !laplace || !fdm)
    return ASM_ERR_INVALID_VALUE;
  *laplace = ctx->pair_a ? 1 : 0;
  *fdm     = ctx->pair_p ? 1 : 0;
  return ASM_OK;
}

int
asm_pair_zchunks(const asm_ctx *ctx, int32_t *zchunks)
{
  if (!ctx || !zchunks)
    return ASM_ERR_INVALID_VALUE;
  const char *env = std::getenv("ASM_PAIR_ZCHUNKS");
  *zchunks        = env != nullptr ? std::atoi(env) : ctx->pair_zchunks;
  return ASM_OK;
}

int
asm_pair_waves(const asm_ctx *ctx, int32_t *waves)
{
  if (!ctx || !waves)
    return ASM_ERR_INVALID_VALUE;
  *waves = ctx->pair_waves == 2 ? 2 : 4;
  return ASM_OK;
}

// the tiling and work order of the wave-pair sweeps for the block `p` describes (host only, no device needed):
// plan = {tiles x, tiles y, z chunks, layers per chunk, first cell of tile 0 in x, in y}
int
asm_pair_plan(const asm_params *p, int32_t n_cu, int32_t plan[6], int32_t *order, int32_t max_order, int32_t *n_order)
{
  if (p == nullptr || plan == nullptr || n_order == nullptr)
    return ASM_ERR_INVALID_VALUE;
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  if (s.g.p != 4)
    {
      g_create_error = "asm_pair_plan: the wave-pair sweeps exist for degree 4";
      return ASM_ERR_NOT_SUPPORTED;
    }
  int out[6];
  pair_plan(s.g, n_cu, out);
  for (int i = 0; i < 6; ++i)
    plan[i] = out[i];
  std::vector<int> ord;
  *n_order = pair_order(s.g, n_cu, ord);
  if (order != nullptr)
    for (int i = 0; i < std::min<int>(max_order, (int)ord.size()); ++i)
      order[i] = ord[i];
  return ASM_OK;
}

// the tiles of that tiling: tiles[i] = {first cell x, y, cells in x, in y, bit 0 / 1: owns its first column / row (else: halo),
// last cell column it may own}; *n_tiles entries (at most max_tiles are written)
int
asm_pair_tiles(const asm_params *p, int32_t n_cu, int32_t (*tiles)[6], int32_t max_tiles, int32_t *n_tiles)
{
  if (p == nullptr || n_tiles == nullptr)
    return ASM_ERR_INVALID_VALUE;
  HostSetup s;
  if (!setup_layout(*p, s))
    {
      g_create_error = s.error;
      return ASM_ERR_INVALID_VALUE;
    }
  if (s.g.p != 4)
    {
      g_create_error = "asm_pair_tiles: the wave-pair sweeps exist for degree 4";
      return ASM_ERR_NOT_SUPPORTED;
    }
  std::vector<std::array<int, 6>> t;
  pair_tiles(s.g, n_cu, t);
  *n_tiles = (int32_t)t.size();
  if (tiles != nullptr)
    for (int i = 0; i < std::min<int>(max_tiles, (int)t.size()); ++i)
      for (int k = 0; k < 6; ++k)
        tiles[i][k] = t[i][k];
  return ASM_OK;
}

// which tiled Laplace sweep the context uses: cell rows (waves) per workgroup, 8 = first generation, 9-11 = second
// (measured at creation or fixed by ASM_TILED_ROWS); 0 = the context has no tiled Laplace sweep
int
asm_tiled_laplace_rows(const asm_ctx *ctx, int32_t *rows)
{
  if (!ctx || !rows)
    return ASM_ERR_INVALID_VALUE;
  *rows = ctx->s.laplace_tiled ? (ctx->laplace_rows == 0 ? 8 : ctx->laplace_rows) : 0;
  return ASM_OK;
}

// measurement aid of bench.py ("peak_measured"): the rate of a plain device-to-device copy of two arrays of `bytes` bytes
// each (read + written bytes per second, best of `reps` launches after one warm-up), allocated and freed here
int
asm_measure_copy_bandwidth(asm_ctx *ctx, int64_t bytes, int32_t reps, double *gb_per_s)
{
  const DeviceScope device_scope(ctx);
  if (bytes < (1 << 20) || reps < 1 || gb_per_s == nullptr)
    {
      ctx->error = "asm_measure_copy_bandwidth: at least 1 MiB per array and one repetition";
      return ASM_ERR_INVALID_VALUE;
    }
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, (size_t)bytes) != hipSuccess || hipMalloc(&b, (size_t)bytes) != hipSuccess)
    {
      (void)hipFree(a);
      (void)hipGetLastError();
      ctx->error = "asm_measure_copy_bandwidth: allocation failed";
      return ASM_ERR_DEVICE;
    }
  hipEvent_t e0, e1;
  int        err = ASM_OK;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    err = ASM_ERR_DEVICE;
  float best = 1e30f;
  if (!err)
    {
      (void)hipMemsetAsync(a, 1, (size_t)bytes, ctx->stream);
      (void)hipMemsetAsync(b, 0, (size_t)bytes, ctx->stream);
      for (int r = 0; r <= reps && !err; ++r)
        {
          (void)hipEventRecord(e0, ctx->stream);
          vec_stream_copy(b, a, bytes / 16, ctx->stream);
          (void)hipEventRecord(e1, ctx->stream);
          float ms = 0.f;
          if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
            err = ASM_ERR_DEVICE;
          else if (r > 0 && ms < best)
            best = ms;
        }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
  (void)hipFree(a);
  (void)hipFree(b);
  if (err)
    {
      ctx->error = "asm_measure_copy_bandwidth: device error";
      return err;
    }
  *gb_per_s = 2.0 * (double)(bytes / 16 * 16) / (best * 1e-3) / 1e9;
  return ASM_OK;
}

#ifdef ASM_STAMP
// diagnostic builds only (not declared in the public header)
int
asm_debug_stamps(asm_ctx *ctx, unsigned long long out[32], int reset)
{
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  asmamd::tiled_debug_stamps(out, reset != 0);
  return ASM_OK;
}
#endif

// ---- testing aids: all ranks of a partition as contexts of ONE process on ONE device -------------
// (a box with one GPU cannot run RCCL between two ranks; these two entry points let the tests run
// the partitioned kernels, layouts and exchange plans on real device buffers all the same)
int
asm_debug_external_exchange(asm_ctx *ctx, int on)
{
  ctx->external_exchange = on != 0;
  return ASM_OK;
}

// the tiled sweeps as two launches (workgroups next to interface faces, then the others), as the
// overlapped exchange launches them, but without an exchange in between
int
asm_debug_two_phase(asm_ctx *ctx, int on)
{
  ctx->debug_two_phase = on != 0;
  return ASM_OK;
}

// the ghost exchange (add == 0: update_ghost_values, add != 0: compress(add)) of vecs[r] of ctxs[r],
// r = rank, for all ranks at once; device-to-device copies take the place of ncclSend/ncclRecv
int
asm_debug_exchange_all(asm_ctx **ctxs, double **vecs, int n, int add)
{
  for (int r = 0; r < n; ++r)
    if (ctxs[r]->s.prm.rank != r || ctxs[r]->s.prm.world_size != n)
      {
        ctxs[r]->error = "asm_debug_exchange_all: ctxs[r] must be rank r of n";
        return ASM_ERR_INVALID_VALUE;
      }
  static const bool dimension_ordered = std::getenv("ASM_COMPRESS_BY_DIMENSION") != nullptr;
  if (add && ctxs[0]->n_dmsg > 0 && !dimension_ordered)
    {
      // single-round compress(add): the ghost box rank r holds for offset o is added to the box its peer sends
      // towards -o in the update
      for (int r = 0; r < n; ++r)
        BOX_TRY_CTX(ctxs[r], vec_boxes(ctxs[r]->g, 0, vecs[r], ctxs[r]->dbuf[1], ctxs[r]->drecv, ctxs[r]->stream));
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
      for (int r = 0; r < n; ++r)
        for (int i = 0; i < ctxs[r]->n_dmsg; ++i)
          {
            const asm_direct_message &m = ctxs[r]->dmsg[i];
            if (m.send_count == 0)
              continue;
            const asm_ctx *           pc = ctxs[m.peer];
            const asm_direct_message *pm = nullptr;
            for (int j = 0; j < pc->n_dmsg; ++j)
              if (pc->dmsg[j].peer == r && pc->dmsg[j].offset[0] == -m.offset[0] && pc->dmsg[j].offset[1] == -m.offset[1] &&
                  pc->dmsg[j].offset[2] == -m.offset[2])
                pm = &pc->dmsg[j];
            if (pm == nullptr || pm->recv_count != m.send_count)
              {
                ctxs[r]->error = "asm_debug_exchange_all: send and receive boxes do not match";
                return ASM_ERR_INVALID_VALUE;
              }
            HIP_TRY_CTX(ctxs[r], hipMemcpy(ctxs[r]->dbuf[0] + m.send_offset, pc->dbuf[1] + pm->recv_offset,
                                           m.send_count * sizeof(double), hipMemcpyDeviceToDevice));
          }
      for (int r = 0; r < n; ++r)
        {
          BOX_TRY_CTX(ctxs[r], vec_boxes(ctxs[r]->g, 2, vecs[r], ctxs[r]->dbuf[0], ctxs[r]->dsend, ctxs[r]->stream));
          BOX_TRY_CTX(ctxs[r], vec_boxes(ctxs[r]->g, 3, vecs[r], nullptr, ctxs[r]->drecv, ctxs[r]->stream));
        }
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
      return ASM_OK;
    }
  if (!add && ctxs[0]->n_dmsg > 0)
    {
      // single-round plan: what rank r receives from offset o is what its peer sends towards -o
      for (int r = 0; r < n; ++r)
        BOX_TRY_CTX(ctxs[r], vec_boxes(ctxs[r]->g, 0, vecs[r], ctxs[r]->dbuf[0], ctxs[r]->dsend, ctxs[r]->stream));
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
      for (int r = 0; r < n; ++r)
        for (int i = 0; i < ctxs[r]->n_dmsg; ++i)
          {
            const asm_direct_message &m = ctxs[r]->dmsg[i];
            if (m.recv_count == 0)
              continue;
            const asm_ctx *           pc = ctxs[m.peer];
            const asm_direct_message *pm = nullptr;
            for (int j = 0; j < pc->n_dmsg; ++j)
              if (pc->dmsg[j].peer == r && pc->dmsg[j].offset[0] == -m.offset[0] && pc->dmsg[j].offset[1] == -m.offset[1] &&
                  pc->dmsg[j].offset[2] == -m.offset[2])
                pm = &pc->dmsg[j];
            if (pm == nullptr || pm->send_count != m.recv_count)
              {
                ctxs[r]->error = "asm_debug_exchange_all: send and receive boxes do not match";
                return ASM_ERR_INVALID_VALUE;
              }
            HIP_TRY_CTX(ctxs[r], hipMemcpy(ctxs[r]->dbuf[1] + m.recv_offset, pc->dbuf[0] + pm->send_offset,
                                           m.recv_count * sizeof(double), hipMemcpyDeviceToDevice));
          }
      for (int r = 0; r < n; ++r)
        BOX_TRY_CTX(ctxs[r], vec_boxes(ctxs[r]->g, 1, vecs[r], ctxs[r]->dbuf[1], ctxs[r]->drecv, ctxs[r]->stream));
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
      return ASM_OK;
    }
  std::vector<std::array<asm_exchange_step, 3>> steps(n);
  int                                           n_steps = 0;
  for (int r = 0; r < n; ++r)
    n_steps = build_exchange_plan(ctxs[r]->s, add != 0, steps[r].data());
  const auto count = [](const int32_t sz[3]) { return (long long)sz[0] * sz[1] * sz[2]; };
  for (int i = 0; i < n_steps; ++i)
    {
      for (int r = 0; r < n; ++r)
        for (int sd = 0; sd < 2; ++sd)
          if (steps[r][i].peer[sd] >= 0 && count(steps[r][i].send_size[sd]) > 0)
            BOX_TRY_CTX(ctxs[r], vec_box(ctxs[r]->g, 0, vecs[r], ctxs[r]->hbuf[sd], steps[r][i].send_lo[sd], steps[r][i].send_size[sd], ctxs[r]->stream));
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
      for (int r = 0; r < n; ++r)
        for (int sd = 0; sd < 2; ++sd)
          {
            const asm_exchange_step &st  = steps[r][i];
            const long long          cnt = st.peer[sd] < 0 ? 0 : count(st.recv_size[sd]);
            if (cnt == 0)
              continue;
            // my low neighbour sends me its high-side slab and vice versa
            const asm_exchange_step &ps = steps[st.peer[sd]][i];
            if (ps.peer[1 - sd] != r || count(ps.send_size[1 - sd]) != cnt)
              {
                ctxs[r]->error = "asm_debug_exchange_all: send and receive slabs do not match";
                return ASM_ERR_INVALID_VALUE;
              }
            HIP_TRY_CTX(ctxs[r], hipMemcpy(ctxs[r]->hbuf[2 + sd], ctxs[st.peer[sd]]->hbuf[1 - sd], cnt * sizeof(double),
                                           hipMemcpyDeviceToDevice));
          }
      for (int r = 0; r < n; ++r)
        {
          const asm_exchange_step &st = steps[r][i];
          for (int sd = 0; sd < 2; ++sd)
            if (st.peer[sd] >= 0 && count(st.recv_size[sd]) > 0)
              BOX_TRY_CTX(ctxs[r], vec_box(ctxs[r]->g, add ? 2 : 1, vecs[r], ctxs[r]->hbuf[2 + sd], st.recv_lo[sd], st.recv_size[sd], ctxs[r]->stream));
          if (add)
            for (int sd = 0; sd < 2; ++sd)
              if (st.peer[sd] >= 0 && count(st.send_size[sd]) > 0)
                BOX_TRY_CTX(ctxs[r], vec_box(ctxs[r]->g, 3, vecs[r], nullptr, st.send_lo[sd], st.send_size[sd], ctxs[r]->stream));
        }
      HIP_TRY_CTX(ctxs[0], hipDeviceSynchronize());
    }
  return ASM_OK;
}

int
asm_time_calls(asm_ctx *ctx, int what, int reps, double *dst, const double *src, float *ms)
{
  const DeviceScope device_scope(ctx);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, ctx->stream));
  int err = ASM_OK;
  for (int i = 0; i < reps && !err; ++i)
    switch (what)
      {
        case 0: err = asm_op_vmult(ctx, dst, src); break;
        case 1: err = asm_precond_vmult(ctx, dst, src); break;
        case 2: err = asm_cheb_vmult(ctx, dst, src); break;
        case 3: err = asm_cheb_step(ctx, dst, src); break;
        default: ctx->error = "asm_time_calls: unknown entry point"; err = ASM_ERR_INVALID_VALUE;
      }
  HIP_TRY(hipEventRecord(e1, ctx->stream));
  HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err;
}
} // extern "C"
