// A hand-written Chebyshev iteration (first-kind polynomial, PreconditionChebyshev::step) that reaches the
// fused, atomics-free sweeps through the public vmult(dst, src, FusedUpdate) surface - what an external
// relaxation / multigrid driver would do - must give bitwise the result of the library's own step().
// Exit code 0: identical; 77: no HIP device; 1: mismatch or error.
#include "dealii_asm_amd.hpp"

#include <cstdio>
#include <cstring>
#include <memory>
#include <random>

int main()
{
  asm_params prm;
  asm_params_default(&prm);
  asm_params_set(&prm, "fe degree", "4");
  asm_params_set(&prm, "degree", "3");
  asm_params_set(&prm, "weighting type", "post");
  prm.n_cells[0] = 13;
  prm.n_cells[1] = 9;
  prm.n_cells[2] = 5;
  if (asm_params_set(&prm, "polynomial type", "5th kind") == ASM_OK || asm_params_set(&prm, "no such key", "1") == ASM_OK)
    return 1; // unknown values / keys must be refused
  try
    {
      asmamd::Context                   ctx(prm);
      asmamd::LaplaceOperatorMatrixFree op(ctx);
      asmamd::ASPoissonPreconditioner   fdm(ctx);
      asmamd::PreconditionChebyshev     smoother(ctx);
      // the drivers hold preconditioners through the virtual base
      const asmamd::PreconditionerBase<asmamd::DeviceVector> *base = &smoother;
      smoother.set_eigenvalue(1.7);
      double theta, delta, lambda_max;
      smoother.parameters(theta, delta, lambda_max);

      asmamd::DeviceVector b, x_ref, x, x_old, r;
      for (asmamd::DeviceVector *v : {&b, &x_ref, &x, &x_old, &r})
        op.initialize_dof_vector(*v);
      std::mt19937                           gen(42);
      std::uniform_real_distribution<double> dist(-1., 1.);
      std::vector<double>                    hb(b.locally_owned_size()), hx(b.locally_owned_size());
      for (auto &v : hb)
        v = dist(gen);
      for (auto &v : hx)
        v = dist(gen);
      b.copy_from_host(hb);
      x_ref.copy_from_host(hx);
      x.copy_from_host(hx);
      base->step(x_ref, b); // the library's own iteration (degree 3)

      // the same iteration through the fused overloads: x_1 = x_0 + 1/theta P^-1 (b - A x_0), then the three-term recurrence
      op.vmult(r, x, asmamd::FusedUpdate::residual(b));
      fdm.vmult(x_old, r, asmamd::FusedUpdate::relax(x, 1., 1. / theta)); // x_old <- x_1 ; x holds x_0
      std::swap(x, x_old);                                                // x = x_1, x_old = x_0
      double       rhok  = delta / theta;
      const double sigma = theta / delta;
      for (int k = 1; k < 3; ++k)
        {
          const double rhokp = 1. / (2. * sigma - rhok);
          const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
          rhok = rhokp;
          op.vmult(r, x, asmamd::FusedUpdate::residual(b));
          fdm.vmult(x_old, r, asmamd::FusedUpdate::cheb_next(x, x_old, f1, f2)); // x_old <- x_{k+1}
          std::swap(x, x_old);
        }
      const auto got = x.copy_to_host(), ref = x_ref.copy_to_host();
      const bool same = std::memcmp(got.data(), ref.data(), got.size() * sizeof(double)) == 0;
      std::printf("hand-written fused Chebyshev loop vs step(): %s, symmetric=%d, n_owned=%lld\n", same ? "bitwise identical" : "DIFFERENT",
                  (int)fdm.is_symmetric(), (long long)op.get_partitioner().n_owned);
      return same ? 0 : 1;
    }
  catch (const std::exception &e)
    {
      std::printf("exception: %s\n", e.what());
      return std::strstr(e.what(), "no HIP device") ? 77 : 1;
    }
}
