// Structured block of the hex mesh owned by one GPU, and the lexicographic box
// layout of its DoF vectors in HBM.
//
// This replaces deal.II's DoFHandler / Utilities::MPI::Partitioner as used by
// the reference (include/matrix_free.h:154-213, include/operator.h:780-849) and
// the 27-start-index compression of include/vector_access_reduced.h: on a
// structured mesh every patch of the reference (cell, cell + overlap layers,
// vertex star; include/dof_tools.h:10-137, 206-300) is a contiguous BOX of the
// lexicographic DoF grid, so no index lists are stored at all.
#pragma once

#include <cstdint>

#ifdef __HIPCC__
#  define ASM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#  define ASM_HD inline
#endif

namespace asmamd
{
  // POD, passed by value to kernels
  struct DevGrid
  {
    int p;  // polynomial degree
    int n1; // p + 1
    int nc[3]; // local cells per direction
    // local logical DoF coordinate I of direction d: cell c covers [p c, p c + p]
    int wrap[3];     // 1: periodic with a single rank in this direction -> I mod (nc p)
    int ext_lo[3];   // smallest / largest logical coordinate that is stored
    int ext_hi[3];
    int dir_lo[3];   // 1: plane 0 / plane nc*p is a homogeneous Dirichlet boundary
    int dir_hi[3];
    int own_hi[3];   // last owned plane (owned range is [0, own_hi])
    long long stride[3]; // element strides of the stored box (x fastest)

    ASM_HD int n_planes(const int d) const { return nc[d] * p; }

    // storage index along direction d of logical coordinate I (must exist)
    ASM_HD int store_index(const int d, int I) const
    {
      if (wrap[d])
        {
          const int n = nc[d] * p;
          I           = I % n;
          return I < 0 ? I + n : I;
        }
      return I - ext_lo[d];
    }
    // same for -n <= I < 2n (anything one patch away from the block): no integer division
    ASM_HD int store_index_near(const int d, const int I) const
    {
      if (wrap[d])
        {
          const int n = nc[d] * p;
          return I < 0 ? I + n : (I >= n ? I - n : I);
        }
      return I - ext_lo[d];
    }
    ASM_HD bool exists(const int d, const int I) const
    {
      return wrap[d] || (I >= ext_lo[d] && I <= ext_hi[d]);
    }
    ASM_HD bool dirichlet(const int d, const int I) const
    {
      return (dir_lo[d] && I == 0) || (dir_hi[d] && I == nc[d] * p);
    }
    ASM_HD long long offset(const int I, const int J, const int K) const
    {
      return store_index(0, I) * stride[0] + store_index(1, J) * stride[1] +
             store_index(2, K) * stride[2];
    }
  };

  // Blocks b and b + 8 of a launch are observed to land on the same XCD (own L2).  xcd_block maps the hardware block
  // index to a work index such that every XCD takes a contiguous range of the work (a slab of the mesh): neighbouring
  // cells / DoF rows, which share cache lines, then meet in one L2.  Placement only, never results.
  ASM_HD long long xcd_block(const long long b, const long long n)
  {
    const long long xcd = b & 7, idx = b >> 3;
    return xcd * (n >> 3) + (xcd < (n & 7) ? xcd : (n & 7)) + idx;
  }

  // cells per column of the accumulate form of the 7^3 sweep: patches per wave along x, waves per workgroup along y
  constexpr int S7_TILE_X = 9, S7_TILE_Y = 4;

  enum PatchKind
  {
    PATCH_CELL   = 0, // element-centred, m = p - 1 + 2 n_overlap (matrix_free.h:90)
    PATCH_VERTEX = 1  // vertex-star, m = 2p - 1 (matrix_free.h:91)
  };

  // per-direction tables of the FDM preconditioner (device pointers)
  struct DevFdm
  {
    int m;          // patch size per direction
    int origin_mul; // patch origin = origin_mul * c + origin_add
    int origin_add;
    int vertex;     // PATCH_VERTEX: patch invalid unless cell c+1 exists in all directions
    int tensor_inst;       // 1: instance id depends on (d, c_d) only
    const int *inst[3];    // [nc[d]] (tensor) or [ncells] (per cell): instance id of direction d
    const double *eig;     // [n_inst][m*m + m]: Q (row i, col a) then lambda
    const int *   eig_sym; // [n_inst] 1: the first (m+1)/2 eigenvectors are even, the others odd vectors
    int           n_inst;
    int           has_pre, has_post; // weights differ from the 0/1 validity mask
    // tensor-product meshes: 1/(lx+ly+lz) for every instance triple, [iz][iy][ix][m^3], or nullptr
    const double *inv_table;
    const double *w_pre[3];  // [nc[d]][m] 1-D factors of the pre weight, or nullptr
    const double *w_post[3]; // [nc[d]][m] 1-D factors of the post weight, or nullptr
    // m = 7 on tensor-product meshes: the box of cells [s7_lo, s7_hi) that share one parity-sorted instance per
    // direction is solved by the slice-per-lane kernel (kernels_slice7.hip), the generic kernel skips it
    int           s7_on; // 0: no; 1: one instance triple for the whole box (tensor meshes); 2: per-cell instances (deformed meshes)
    int           s7_lo[3], s7_hi[3], s7_inst[3];
    const double *s7_inv;    // [a][c][b] (x, z, y eigen-index) reciprocal eigenvalue sums of that instance triple
    const double *s7_wyz[2]; // [k][j] products of the z and y weight factors of the box's cells (pre, post)
    const int *   s7_others; // the cells outside the box (linear index), for the generic kernel
    int           s7_n_others;
    // accumulate form (fdm_slice7_acc): flag[d][storage index] = 0 where exactly one column (S7_TILE_X x S7_TILE_Y cells) /
    // z chunk (s7_lz layers) of the box touches the DoF plane and no cell outside the box does; nullptr: not available
    const unsigned char *s7_flag[3];
    int                  s7_lz;
    int           small;     // m <= 3: one thread per cell, the patch in registers (kernels_small.hip)
  };

  // Tables of a "Kronecker-diagonal" cell operator  sum_cells R^T (Bz x By x Bx) diag(D) (Fz x Fy x Fx)^T R  on a
  // tensor-product mesh, as the wave-pair sweep (kernels_pair.hip) applies it: the FDM preconditioner (F = diag(w_pre) Q,
  // B = diag(w_post) Q, D = 1 / (lx + ly + lz)) and the Laplace operator on box cells (F = constrained rows of F_ref zeroed,
  // B = F_ref, D = hy hz / hx la + hx hz / hy lb + hx hy / hz lc) differ in these tables only.  A cell's class per
  // direction selects its matrices; the class triple selects the diagonal.  Device pointers.
  constexpr int KRON_MAX_CLS    = 4;   // classes per direction
  constexpr int KRON_MAT_STRIDE = 96;  // doubles per (direction, class): F^T [a][6] 30 (column a of F, padded), B [i][6] 30, F even-odd 16, B even-odd 16
  constexpr int KRON_DIAG_STRIDE = 128; // doubles per class triple: D[(b * 5 + a) * 5 + c]
  struct DevKron
  {
    int           valid;
    int           mask_constrained; // 1: constrained (Dirichlet) DoFs of the source are read as zero (Laplace operator)
    int           n_cls[3];
    const int *   cls[3]; // [nc[d] + 2]: class of cell c at [c + 1]
    const int *   sym;    // [3][KRON_MAX_CLS] 1: the columns 0..2 of F and B are even, 3..4 odd vectors
    const double *mat;    // [3][KRON_MAX_CLS][KRON_MAT_STRIDE]
    const double *diag;   // [clz][cly][clx][KRON_DIAG_STRIDE], KRON_MAX_CLS classes per direction
  };

  // geometry + basis of the Laplace operator (device pointers)
  struct DevOp
  {
    int cartesian;        // 1: h[d][c_d] cell sizes; 0: 27 monomial coefficient vectors per cell
    const double *h[3];   // [nc[d]]
    const double *coeff;  // [ncells][27][3] (operator.h:625-661 layout: index 9k+3j+i, component)
    const double *N;      // [q][i]
    const double *Dc;     // [q][r]
    const double *xq;     // [q]
    const double *wq;     // [q]
    // degree 4: M_ref = F F^T, K_ref = F diag(lam) F^T with parity-sorted columns of F (numerics.h: Basis1D::diagonalise)
    const double *F_ref;   // [i][a]
    const double *lam_ref; // [a]
    // degree <= 2 on Cartesian cells: one thread per cell with the 1-D reference mass and stiffness matrices
    // ([i][j], row length p + 1) as kernel arguments (kernels_small.hip)
    int    small;
    double M_small[9], K_small[9];
  };
} // namespace asmamd
