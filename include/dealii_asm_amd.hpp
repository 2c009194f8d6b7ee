// dealii_asm_amd.hpp -- header-only C++ shim over the C ABI (dealii_asm_amd.h) that
// re-exposes the operator surface of the reference so that a deal.II-style
// Chebyshev / multigrid driver can hold these objects where it held the reference's:
//
//   asmamd::DeviceVector               <-> LinearAlgebra::distributed::Vector<double>
//   asmamd::LaplaceOperatorMatrixFree  <-> LaplaceOperatorMatrixFree      (include/operator.h:266-1628)
//   asmamd::ASPoissonPreconditioner    <-> ASPoissonPreconditioner        (include/matrix_free.h:63-1568)
//   asmamd::PreconditionChebyshev      <-> PreconditionChebyshev<Op,Vec,P> as configured by
//                                          create_chebyshev_preconditioner (include/precondition.templates.h:89-158),
//                                          with the PreconditionerBase vmult/step shape
//                                          (include/preconditioners.h:725-742)
//
// The one place where the surface is narrowed: the reference's
// vmult(dst, src, pre, post) overloads take host lambdas over index ranges
// (include/operator.h:1367-1430, include/matrix_free.h:960-986).  Host code cannot
// run inside a GPU sweep, so the fused vector updates PreconditionChebyshev passes
// through those lambdas are applied by the kernels' store epilogues instead and
// PreconditionChebyshev::vmult/step are the entry points that use them.
// Errors: the reference throws deal.II exceptions; this shim throws std::runtime_error.
#ifndef DEALII_ASM_AMD_HPP
#define DEALII_ASM_AMD_HPP

#include "dealii_asm_amd.h"

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace asmamd
{
  class Context
  {
  public:
    explicit Context(const asm_params &params)
    {
      asm_ctx *h = nullptr;
      if (const int err = asm_create(&params, &h))
        throw std::runtime_error("asm_create failed (" + std::to_string(err) + "): " + asm_last_error(nullptr));
      handle_.reset(h, asm_destroy);
      check(asm_vector_layout(h, &layout_));
    }
    // a level of a multigrid hierarchy: owned by the asm_mg object, not by this wrapper
    Context(asm_ctx *borrowed, std::nullptr_t)
      : handle_(borrowed, [](asm_ctx *) {})
    {
      check(asm_vector_layout(borrowed, &layout_));
    }
    asm_ctx *get() const { return handle_.get(); }
    const asm_layout &layout() const { return layout_; }
    void check(const int err) const
    {
      if (err)
        throw std::runtime_error(std::string("dealii_asm_amd: ") + asm_last_error(handle_.get()));
    }
    void synchronize() const { check(asm_synchronize(handle_.get())); }
    // promise that vectors are modified between calls only through this library (asm_set_keep_ghosts)
    void set_keep_ghosts(const bool on) const { check(asm_set_keep_ghosts(handle_.get(), on ? 1 : 0)); }

  private:
    std::shared_ptr<asm_ctx> handle_;
    asm_layout               layout_;
  };

  // device vector in the library's box layout; initialize_dof_vector() creates it
  class DeviceVector
  {
  public:
    DeviceVector() = default;
    DeviceVector(const DeviceVector &) = delete;
    DeviceVector &operator=(const DeviceVector &) = delete;
    DeviceVector(DeviceVector &&o) noexcept { swap(o); }
    DeviceVector &operator=(DeviceVector &&o) noexcept
    {
      swap(o);
      return *this;
    }
    ~DeviceVector() { clear(); }

    void reinit(const Context &ctx)
    {
      clear();
      ctx_ = &ctx;
      ctx.check(asm_vector_alloc(ctx.get(), &data_));
    }
    double *      data() { return data_; }
    const double *data() const { return data_; }
    std::size_t   locally_owned_size() const { return ctx_ ? ctx_->layout().n_owned : 0; }
    // host copies of the locally owned range, lexicographic (x fastest)
    void copy_from_host(const std::vector<double> &owned)
    {
      if (owned.size() != locally_owned_size())
        throw std::runtime_error("copy_from_host: size mismatch");
      ctx_->check(asm_vector_import(ctx_->get(), data_, owned.data()));
    }
    std::vector<double> copy_to_host() const
    {
      std::vector<double> out(locally_owned_size());
      ctx_->check(asm_vector_export(ctx_->get(), out.data(), data_));
      return out;
    }
    void update_ghost_values() { ctx_->check(asm_update_ghost_values(ctx_->get(), data_)); }
    void compress_add() { ctx_->check(asm_compress_add(ctx_->get(), data_)); }

  private:
    void clear()
    {
      if (data_ && ctx_)
        asm_vector_free(ctx_->get(), data_);
      data_ = nullptr;
    }
    void swap(DeviceVector &o)
    {
      std::swap(ctx_, o.ctx_);
      std::swap(data_, o.data_);
    }
    const Context *ctx_  = nullptr;
    double *       data_ = nullptr;
  };

  // PreconditionerBase<VectorType>, include/preconditioners.h:725-742: vmult is pure virtual, step throws
  // unless overridden.  A multigrid smoother wrapper (WrapperForGMG, include/precondition.h:24-78) or a
  // Krylov solver holds a pointer to this base.
  template <typename VectorType>
  class PreconditionerBase
  {
  public:
    using vector_type = VectorType;
    virtual ~PreconditionerBase() = default;
    virtual void vmult(VectorType &dst, const VectorType &src) const = 0;
    virtual void step(VectorType &, const VectorType &) const { throw std::runtime_error("step: not implemented (ExcNotImplemented)"); }
  };

  // The fused vector update a caller hands to vmult(dst, src, update) where the reference's drivers hand
  // pre/post lambdas (include/operator.h:1367-1430, include/matrix_free.h:960-986): a closed set
  // (asm_fused_update) that the sweeps apply in their store phase.
  struct FusedUpdate
  {
    int                 mode  = ASM_FUSED_NONE;
    const DeviceVector *b     = nullptr; // ASM_FUSED_RESIDUAL
    const DeviceVector *x     = nullptr; // ASM_FUSED_RELAX, ASM_FUSED_CHEB_NEXT
    const DeviceVector *x_old = nullptr; // ASM_FUSED_CHEB_NEXT (dst may be the same vector)
    double              f1 = 0., f2 = 1.;
    static FusedUpdate residual(const DeviceVector &b)
    {
      FusedUpdate u;
      u.mode = ASM_FUSED_RESIDUAL;
      u.b    = &b;
      return u;
    }
    static FusedUpdate cheb_first(const double f2)
    {
      FusedUpdate u;
      u.mode = ASM_FUSED_CHEB_FIRST;
      u.f2   = f2;
      return u;
    }
    static FusedUpdate relax(const DeviceVector &x, const double f1, const double f2)
    {
      FusedUpdate u;
      u.mode = ASM_FUSED_RELAX;
      u.x    = &x;
      u.f1   = f1;
      u.f2   = f2;
      return u;
    }
    static FusedUpdate cheb_next(const DeviceVector &x, const DeviceVector &x_old, const double f1, const double f2)
    {
      FusedUpdate u;
      u.mode  = ASM_FUSED_CHEB_NEXT;
      u.x     = &x;
      u.x_old = &x_old;
      u.f1    = f1;
      u.f2    = f2;
      return u;
    }
  };

  // operator concept used by Chebyshev / Krylov / mg::Matrix (include/operator.h:273-283, 755-778, 1353-1524)
  class LaplaceOperatorMatrixFree
  {
  public:
    using value_type  = double;
    using vector_type = DeviceVector;
    static constexpr int dimension = 3;
    explicit LaplaceOperatorMatrixFree(const Context &ctx)
      : ctx_(ctx)
    {}
    void vmult(DeviceVector &dst, const DeviceVector &src) const { ctx_.check(asm_op_vmult(ctx_.get(), dst.data(), src.data())); }
    // vmult(dst, src, pre, post), include/operator.h:1367-1430
    void vmult(DeviceVector &dst, const DeviceVector &src, const FusedUpdate &u) const
    {
      ctx_.check(asm_op_vmult_fused(ctx_.get(), u.mode, dst.data(), src.data(), u.b ? u.b->data() : nullptr));
    }
    void Tvmult(DeviceVector &, const DeviceVector &) const { throw std::runtime_error("Tvmult: not implemented (as in the reference)"); }
    // the vector layout stands in for the Utilities::MPI::Partitioner (include/operator.h:1491-1498)
    const asm_layout &get_partitioner() const { return ctx_.layout(); }
    void initialize_dof_vector(DeviceVector &v) const { v.reinit(ctx_); }
    long long m() const { return ctx_.layout().n_global_dofs; }
    void compute_inverse_diagonal(DeviceVector &diag) const
    {
      diag.reinit(ctx_);
      ctx_.check(asm_op_inverse_diagonal(ctx_.get(), diag.data()));
    }
    static constexpr bool is_matrix_free() { return true; }

  private:
    const Context &ctx_;
  };

  // ASPoissonPreconditioner behind the PreconditionerBase shape (include/preconditioners.h:725-742, 897-926)
  class ASPoissonPreconditioner : public PreconditionerBase<DeviceVector>
  {
  public:
    explicit ASPoissonPreconditioner(const Context &ctx)
      : ctx_(ctx)
    {}
    void vmult(DeviceVector &dst, const DeviceVector &src) const override
    {
      ctx_.check(asm_precond_vmult(ctx_.get(), dst.data(), src.data()));
    }
    // vmult(dst, src, pre, post), include/matrix_free.h:960-986
    void vmult(DeviceVector &dst, const DeviceVector &src, const FusedUpdate &u) const
    {
      ctx_.check(asm_precond_vmult_fused(ctx_.get(), u.mode, dst.data(), src.data(), u.x ? u.x->data() : nullptr,
                                         u.x_old ? u.x_old->data() : nullptr, u.f1, u.f2));
    }
    // include/matrix_free.h:896-904
    bool is_symmetric() const
    {
      int32_t s = 0;
      ctx_.check(asm_is_symmetric(ctx_.get(), &s));
      return s != 0;
    }
    const asm_layout &get_partitioner() const { return ctx_.layout(); } // include/matrix_free.h:994-998
    long long n_fdm_instances() const
    {
      int64_t n = 0;
      ctx_.check(asm_n_fdm_instances(ctx_.get(), &n));
      return n;
    }
    std::size_t memory_consumption() const
    {
      int64_t n = 0;
      ctx_.check(asm_memory_consumption(ctx_.get(), &n));
      return n;
    }

  private:
    const Context &ctx_;
  };

  class PreconditionChebyshev : public PreconditionerBase<DeviceVector>
  {
  public:
    struct EigenvalueInformation
    {
      double min_eigenvalue_estimate, max_eigenvalue_estimate;
    };
    explicit PreconditionChebyshev(const Context &ctx)
      : ctx_(ctx)
    {}
    // estimate_eigenvalues as driven by include/precondition.templates.h:458-463
    EigenvalueInformation estimate_eigenvalues() const
    {
      double lam = 0.;
      ctx_.check(asm_cheb_estimate_eigenvalues(ctx_.get(), &lam));
      return {lam, 1.2 * lam};
    }
    void set_eigenvalue(const double lambda_max) const { ctx_.check(asm_cheb_set_eigenvalue(ctx_.get(), lambda_max)); }
    void vmult(DeviceVector &dst, const DeviceVector &src) const override
    {
      ctx_.check(asm_cheb_vmult(ctx_.get(), dst.data(), src.data()));
    }
    void step(DeviceVector &dst, const DeviceVector &src) const override
    {
      ctx_.check(asm_cheb_step(ctx_.get(), dst.data(), src.data()));
    }
    // theta, delta, lambda_max of the polynomial in use
    void parameters(double &theta, double &delta, double &lambda_max) const
    {
      double out[3];
      ctx_.check(asm_cheb_parameters(ctx_.get(), out));
      theta      = out[0];
      delta      = out[1];
      lambda_max = out[2];
    }

  private:
    const Context &ctx_;
  };

  // ---- the smoother's callers -------------------------------------------------------------------------
  // PreconditionerGMG / MyMultigrid (include/multigrid.h:40-520, include/precondition.h:82-186): level
  // operators, smoothers ("mg smoother" = the smoother the fine parameters describe), MGTwoLevelTransfer
  // between neighbouring levels, coarse-grid solver, one V-cycle per vmult
  class PreconditionerGMG : public PreconditionerBase<DeviceVector>
  {
  public:
    PreconditionerGMG(const asm_params &fine, const asm_mg_params &mp)
    {
      asm_mg *h = nullptr;
      if (const int err = asm_mg_create(&fine, &mp, &h))
        throw std::runtime_error("asm_mg_create failed (" + std::to_string(err) + "): " + asm_mg_last_error(nullptr));
      handle_.reset(h, asm_mg_destroy);
      int32_t n = 0;
      asm_mg_n_levels(h, &n);
      for (int32_t l = 0; l < n; ++l)
        {
          asm_ctx *c = nullptr;
          check(asm_mg_level_context(h, l, &c));
          levels_.emplace_back(new Context(c, nullptr));
        }
    }
    asm_mg *       get() const { return handle_.get(); }
    unsigned int   n_levels() const { return levels_.size(); }
    const Context &level(const unsigned int l) const { return *levels_[l]; }   // mg_operators[l]
    const Context &fine() const { return *levels_.back(); }
    void vmult(DeviceVector &dst, const DeviceVector &src) const override { check(asm_mg_vmult(handle_.get(), dst.data(), src.data())); }
    // MGTwoLevelTransfer between level and level - 1 (include/multigrid.h:289-299, 336-346)
    void prolongate_and_add(const unsigned int level, DeviceVector &dst_fine, const DeviceVector &src_coarse) const
    {
      check(asm_mg_prolongate_add(handle_.get(), (int32_t)level, dst_fine.data(), src_coarse.data()));
    }
    void restrict_and_add(const unsigned int level, DeviceVector &dst_coarse_zeroed, const DeviceVector &src_fine) const
    {
      check(asm_mg_restrict(handle_.get(), (int32_t)level, dst_coarse_zeroed.data(), src_fine.data()));
    }
    // several ranks: ONE communicator for all block-decomposed levels (collective; id = the bytes of asm_comm_unique_id,
    // broadcast by the launcher), then the levels' eigenvalue estimates; include/multigrid.h:289-346
    void comm_init(const char id[128]) { check(asm_mg_comm_init(handle_.get(), id)); }
    bool level_is_distributed(const unsigned int l) const
    {
      int32_t d = 0;
      check(asm_mg_level_distributed(handle_.get(), (int32_t)l, &d));
      return d != 0;
    }
    // coarse-grid CG solves that stopped at their iteration limit since creation
    long long coarse_failures() const
    {
      int64_t n = 0;
      check(asm_mg_coarse_failures(handle_.get(), &n));
      return n;
    }
    void check(const int err) const
    {
      if (err)
        throw std::runtime_error(std::string("dealii_asm_amd multigrid: ") + asm_mg_last_error(handle_.get()));
    }

  private:
    std::shared_ptr<asm_mg>               handle_;
    std::vector<std::unique_ptr<Context>> levels_;
  };

  // solve(), element_centered_preconditioners_01.cc:108-263: x = 0, SolverCG / SolverGMRES under ReductionControl;
  // returns SolverControl::last_step() and the residuals
  inline asm_solver_result
  solve(const Context &fine, DeviceVector &x, const DeviceVector &b, const int system_preconditioner, const asm_solver_params &sp,
        const PreconditionerGMG *mg = nullptr)
  {
    asm_solver_result res;
    const int err = asm_solve(fine.get(), mg ? mg->get() : nullptr, system_preconditioner, &sp, x.data(), b.data(), &res);
    if (err)
      throw std::runtime_error(std::string("asm_solve: ") + (mg ? asm_mg_last_error(mg->get()) : asm_last_error(fine.get())));
    return res;
  }
} // namespace asmamd

#endif
