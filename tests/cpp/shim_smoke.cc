// Drives the smoother through the C++ shim the way the reference's benchmark does
// (matrix_free_loop_08.likwid.cc:300-347): src = 1, estimate eigenvalues, step().
// Exit code 0: ran on a GPU; 77: no HIP device (the library refused, as it must).
#include "dealii_asm_amd.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

int main()
{
  asm_params prm;
  asm_params_default(&prm);
  prm.degree = 4;
  for (int d = 0; d < 3; ++d)
    prm.n_cells[d] = 6;
  prm.cheb_degree = 3;
  try
    {
      asmamd::Context                   ctx(prm);
      asmamd::LaplaceOperatorMatrixFree op(ctx);
      asmamd::PreconditionChebyshev     smoother(ctx);
      asmamd::DeviceVector              src, dst, res;
      op.initialize_dof_vector(src);
      op.initialize_dof_vector(dst);
      op.initialize_dof_vector(res);
      src.copy_from_host(std::vector<double>(src.locally_owned_size(), 1.0));
      const auto ev = smoother.estimate_eigenvalues();
      std::printf("n dofs %lld, min ev %g max ev %g\n", op.m(), ev.min_eigenvalue_estimate, ev.max_eigenvalue_estimate);
      smoother.vmult(dst, src);
      for (int i = 0; i < 5; ++i)
        smoother.step(dst, src);
      // after a few sweeps A dst is close to src away from the boundary
      op.vmult(res, dst);
      const auto r = res.copy_to_host();
      double     nrm = 0;
      for (const double v : r)
        nrm += (v - 1.0) * (v - 1.0);
      std::printf("|A x - b| / sqrt(n) = %g\n", std::sqrt(nrm / r.size()));
      return std::isfinite(nrm) ? 0 : 1;
    }
  catch (const std::exception &e)
    {
      std::printf("exception: %s\n", e.what());
      return std::strstr(e.what(), "no HIP device") ? 77 : 1;
    }
}
