// A multigrid-preconditioned solve through the C++ shim, shaped like the reference driver
// (element_centered_preconditioners_01.cc:540-748): level hierarchy + V-cycle behind the PreconditionerBase
// pointer a Krylov solver holds, GMRES to a relative tolerance.
// Exit code 0: converged on a GPU; 77: no HIP device; 1: error.
#include "dealii_asm_amd.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

int main()
{
  asm_params prm;
  asm_params_default(&prm);
  asm_params_set(&prm, "fe degree", "4");
  asm_params_set(&prm, "degree", "2");          // Chebyshev degree of the "mg smoother"
  asm_params_set(&prm, "weighting type", "post");
  for (int d = 0; d < 3; ++d)
    prm.n_cells[d] = 8;
  asm_mg_params mp;
  asm_mg_params_default(&mp);
  mp.mg_type = ASM_MG_PH; // "mg type": "ph", "mg p sequence": "bisect"
  try
    {
      asmamd::PreconditionerGMG mg(prm, mp);
      const asmamd::PreconditionerBase<asmamd::DeviceVector> *preconditioner = &mg;
      asmamd::LaplaceOperatorMatrixFree                       op(mg.fine());
      asmamd::DeviceVector                                    b, x, r;
      op.initialize_dof_vector(b);
      op.initialize_dof_vector(x);
      op.initialize_dof_vector(r);
      b.copy_from_host(std::vector<double>(b.locally_owned_size(), 1.0));
      preconditioner->vmult(x, b); // one V-cycle
      asm_solver_params sp;
      asm_solver_params_default(&sp);
      sp.type              = ASM_SOLVER_GMRES;
      sp.rel_tolerance     = 1e-8;
      sp.max_n_tmp_vectors = 17;
      const asm_solver_result res = asmamd::solve(mg.fine(), x, b, ASM_SP_MULTIGRID, sp, &mg);
      std::printf("levels %u, n dofs %lld, GMRES it %d, residual %.3e -> %.3e\n", mg.n_levels(), op.m(), res.n_iterations,
                  res.initial_residual, res.final_residual);
      return (res.converged && res.n_iterations < 20) ? 0 : 1;
    }
  catch (const std::exception &e)
    {
      std::printf("exception: %s\n", e.what());
      return std::strstr(e.what(), "no HIP device") ? 77 : 1;
    }
}
