// Host-side setup of one rank's block: domain decomposition, vector layout,
// geometry, FDM eigen-decompositions and weights.  Pure C++ (no HIP calls), so
// the layout part can be exercised without a GPU (asm_describe_layout).
//
// Follows the setup performed by the reference in
//   ASPoissonPreconditioner::ASPoissonPreconditioner   include/matrix_free.h:73-894
//   LaplaceOperatorMatrixFree::setup_mapping_and_indices include/operator.h:490-753
//   GridTools::compute_harmonic_patch_extend             include/grid_tools.h:54-138
//   TensorProductMatrixCreator::create_laplace_tensor_product_matrix
//                                          include/tensor_product_matrix_creator.h:7-61
// re-designed for a structured mesh: patches are boxes, weights and FDM
// instances are tensor products of 1-D tables.
#pragma once

#include "../../include/dealii_asm_amd.h"
#include "grid.h"
#include "numerics.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace asmamd
{
  enum FaceKind
  {
    FACE_DIRICHLET = 1,
    FACE_NEUMANN   = 2,
    FACE_INTERFACE = 3, // neighbour rank (also a periodic neighbour)
    FACE_WRAP      = 4  // periodic, single rank in this direction
  };

  inline int
  imod(int a, const int n)
  {
    a %= n;
    return a < 0 ? a + n : a;
  }

  // Kershaw map on the unit cube, restated from include/kershaw.h:3-80
  inline double
  k_right(const double eps, const double x)
  {
    return (x <= 0.5) ? (2 - eps) * x : 1 + eps * (x - 1);
  }
  inline double
  k_left(const double eps, const double x)
  {
    return 1 - k_right(eps, 1 - x);
  }
  inline double
  k_step(const double a, const double b, const double x)
  {
    if (x <= 0)
      return a;
    if (x >= 1)
      return b;
    return a + (b - a) * (x * x * x * (x * (6 * x - 15) + 10));
  }
  inline void
  kershaw_map(const double epsy, const double epsz, const double x, const double y, const double z, double *X)
  {
    X[0]                = x;
    const int    layer  = (int)(x * 6.0);
    const double lambda = (x - layer / 6.0) * 6;
    switch (layer)
      {
        case 0:
          X[1] = k_left(epsy, y);
          X[2] = k_left(epsz, z);
          break;
        case 1:
        case 4:
          X[1] = k_step(k_left(epsy, y), k_right(epsy, y), lambda);
          X[2] = k_step(k_left(epsz, z), k_right(epsz, z), lambda);
          break;
        case 2:
          X[1] = k_step(k_right(epsy, y), k_left(epsy, y), lambda / 2);
          X[2] = k_step(k_right(epsz, z), k_left(epsz, z), lambda / 2);
          break;
        case 3:
          X[1] = k_step(k_right(epsy, y), k_left(epsy, y), (1 + lambda) / 2);
          X[2] = k_step(k_right(epsz, z), k_left(epsz, z), (1 + lambda) / 2);
          break;
        default:
          X[1] = k_right(epsy, y);
          X[2] = k_right(epsz, z);
          break;
      }
  }

  struct HostSetup
  {
    asm_params prm;
    int        P[3], pc[3], cell_begin[3], ncg[3];
    int        face[3][2]; // FaceKind of the local block's faces
    int        nbr[3][2];  // neighbour rank per face or -1
    // Hardware test mode (ASM_SELF_INTERFACE, one rank): a periodic direction is closed through ghost planes and
    // messages of the rank to itself instead of index wrap-around, so that the whole multi-rank path (plans, RCCL
    // send/recv, overlap streams, compress(add), all-reduce) runs on a single GPU.
    bool       self_interface = false;
    DevGrid    g;
    int        store_size[3];
    long long  n_alloc, n_owned, n_ghost, n_global;
    int        m;      // patch size 1-D
    int        halo_lo, halo_hi; // ghost planes below plane 0 / from plane nc*p upwards on interface faces
    int        kernel; // asm_kernel_variant in use
    bool       fdm_tiled = false, laplace_tiled = false; // which sweeps of a tiled context run on the tiled kernels

    // extended cell box [-1, nc]^3 (halo cells for owner-computes kernels)
    int ence[3]; // nc + 2
    std::vector<char> cell_exists; // [ext cells]

    // operator data
    bool                cartesian;
    std::vector<double> h[3];  // [nc+2] cell size per direction (Cartesian)
    std::vector<double> coeff; // [ext cells][27][3] monomial coefficients (general)

    // FDM data
    bool                tensor_inst;
    std::vector<int>    inst[3]; // [nc+2] or [ext cells]
    std::vector<double> eig;     // [n_inst][m*m+m]
    std::vector<int>    eig_sym; // [n_inst] 1: eigenvectors sorted by parity (numerics.h: sort_eigenvectors_by_parity)
    int                 n_inst;
    std::vector<double> w_pre[3], w_post[3]; // [nc+2][m]
    bool                has_pre, has_post;

    std::string error;

    int ext_cell(const int cx, const int cy, const int cz) const
    {
      return (cx + 1) + ence[0] * ((cy + 1) + ence[1] * (cz + 1));
    }
  };

  // ---------------------------------------------------------------------
  // decomposition + layout (no geometry)
  // ---------------------------------------------------------------------
  inline bool
  setup_layout(const asm_params &prm_in, HostSetup &s)
  {
    if (prm_in.struct_size != (uint32_t)sizeof(asm_params))
      {
        s.error = "asm_params of another ABI revision (struct_size " + std::to_string(prm_in.struct_size) + ", expected " +
                  std::to_string(sizeof(asm_params)) + "): fill it with asm_params_default first";
        return false;
      }
    s.prm            = prm_in;
    asm_params &prm  = s.prm;
    const int   p    = prm.degree;
    if (p < 1 || p > 7)
      {
        s.error = "degree must be in [1,7] (MAX_DEGREE_RW, include/vector_access_reduced.h:7)";
        return false;
      }
    for (int d = 0; d < 3; ++d)
      {
        if (prm.n_cells[d] < 1)
          {
            s.error = "n_cells must be positive";
            return false;
          }
        const bool per0 = prm.boundary[d][0] == ASM_BC_PERIODIC, per1 = prm.boundary[d][1] == ASM_BC_PERIODIC;
        if (per0 != per1)
          {
            s.error = "periodic boundaries must come in pairs";
            return false;
          }
        for (int f = 0; f < 2; ++f)
          if (prm.boundary[d][f] < 0 || prm.boundary[d][f] > 2)
            {
              s.error = "unknown boundary type";
              return false;
            }
        s.ncg[d] = prm.n_cells[d];
      }
    if (prm.world_size < 1 || prm.rank < 0 || prm.rank >= prm.world_size)
      {
        s.error = "invalid rank / world_size";
        return false;
      }
    // ---- processor grid: split z first, then y, then x (contiguous halos)
    int P[3] = {prm.proc_grid[0], prm.proc_grid[1], prm.proc_grid[2]};
    if (P[0] * P[1] * P[2] != prm.world_size || P[0] < 1 || P[1] < 1 || P[2] < 1)
      {
        P[0] = P[1] = P[2] = 1;
        int rest           = prm.world_size;
        int d              = 2;
        for (int f = 2; rest > 1;)
          {
            if (rest % f == 0)
              {
                P[d] *= f;
                rest /= f;
                d = (d + 2) % 3; // z, y, x, z, ...
              }
            else
              ++f;
          }
      }
    for (int d = 0; d < 3; ++d)
      {
        s.P[d]            = P[d];
        prm.proc_grid[d] = P[d];
        if (P[d] > s.ncg[d])
          {
            s.error = "more ranks than cells in one direction";
            return false;
          }
      }
    s.pc[0] = prm.rank % P[0];
    s.pc[1] = (prm.rank / P[0]) % P[1];
    s.pc[2] = prm.rank / (P[0] * P[1]);

    // ---- Chebyshev parameters (deal.II's smoothing_range <= 1 means "use the smallest eigenvalue of the
    // estimate", which this path does not compute: rejected instead of dividing by it)
    if (prm.cheb_degree < 1)
      {
        s.error = "Chebyshev degree must be >= 1";
        return false;
      }
    if (!(prm.smoothing_range > 1.))
      {
        s.error = "smoothing range must be > 1";
        return false;
      }
    // ---- Chebyshev parameters (deal.II's smoothing_range <= 1 means "use the smallest eigenvalue of the
    // estimate", which this path does not compute: rejected instead of dividing by it)
    if (prm.cheb_degree < 1)
      {
        s.error = "Chebyshev degree must be >= 1";
        return false;
      }
    if (!(prm.smoothing_range > 1.))
      {
        s.error = "smoothing range must be > 1";
        return false;
      }
    // ---- patch shape
    if (prm.n_overlap < 1)
      {
        s.error = "n_overlap must be >= 1";
        return false;
      }
    prm.n_overlap = std::min(prm.n_overlap, p); // precondition.templates.h:195-196
    const bool vertex = prm.element_centric == 0;
    if (vertex && p < 2)
      {
        s.error = "vertex-star patches need degree >= 2";
        return false;
      }
    s.m = vertex ? (2 * p - 1) : (p + 2 * prm.n_overlap - 1);
    if (s.m > 13)
      {
        s.error = "patch size > 13 not supported (EXPAND_OPERATIONS_RWV, include/matrix_free.h:25-36)";
        return false;
      }

    // ---- kernel family
    // the tiled FDM sweep covers degree 4 cell patches with n_overlap 1, the tiled Laplace sweep degree 4
    // on Cartesian cells; a context uses each of them where it applies and the generic kernel otherwise
    s.kernel = prm.kernel_variant;
    const bool fdm_ok = (!vertex && prm.n_overlap == 1 && p == 4), laplace_ok = (p == 4 && prm.geometry == ASM_GEOM_CARTESIAN);
    const bool tiled_ok = fdm_ok || laplace_ok;
    if (s.kernel == ASM_KERNEL_AUTO)
      s.kernel = tiled_ok ? ASM_KERNEL_TILED : ASM_KERNEL_GENERIC;
    if (s.kernel == ASM_KERNEL_TILED && !tiled_ok)
      {
        s.error = "tiled kernels exist for degree 4 only: FDM for n_overlap 1 element-centric patches, Laplace for Cartesian cells";
        return false;
      }
    s.fdm_tiled     = s.kernel == ASM_KERNEL_TILED && fdm_ok;
    s.laplace_tiled = s.kernel == ASM_KERNEL_TILED && laplace_ok;

    s.self_interface = prm.world_size == 1 && std::getenv("ASM_SELF_INTERFACE") != nullptr &&
                       (prm.boundary[0][0] == ASM_BC_PERIODIC || prm.boundary[1][0] == ASM_BC_PERIODIC || prm.boundary[2][0] == ASM_BC_PERIODIC);
    DevGrid &g = s.g;
    g.p        = p;
    g.n1       = p + 1;
    s.n_owned  = 1;
    s.n_global = 1;
    for (int d = 0; d < 3; ++d)
      {
        const bool periodic = prm.boundary[d][0] == ASM_BC_PERIODIC;
        s.cell_begin[d]     = (int)((long long)s.ncg[d] * s.pc[d] / P[d]);
        const int cell_end  = (int)((long long)s.ncg[d] * (s.pc[d] + 1) / P[d]);
        g.nc[d]             = cell_end - s.cell_begin[d];
        for (int f = 0; f < 2; ++f)
          {
            const bool at_global = f == 0 ? (s.pc[d] == 0) : (s.pc[d] == P[d] - 1);
            int        pcn[3]    = {s.pc[0], s.pc[1], s.pc[2]};
            pcn[d]               = imod(s.pc[d] + (f == 0 ? -1 : 1), P[d]);
            const int nbr_rank   = pcn[0] + P[0] * (pcn[1] + P[1] * pcn[2]);
            if (!at_global)
              {
                s.face[d][f] = FACE_INTERFACE;
                s.nbr[d][f]  = nbr_rank;
              }
            else if (periodic)
              {
                const bool wrap = P[d] == 1 && !s.self_interface;
                s.face[d][f]    = wrap ? FACE_WRAP : FACE_INTERFACE;
                s.nbr[d][f]     = wrap ? -1 : nbr_rank;
              }
            else
              {
                s.face[d][f] = prm.boundary[d][f] == ASM_BC_DIRICHLET ? FACE_DIRICHLET : FACE_NEUMANN;
                s.nbr[d][f]  = -1;
              }
          }
        const int np   = g.nc[d] * p;
        g.wrap[d]      = s.face[d][0] == FACE_WRAP;
        g.dir_lo[d]    = s.face[d][0] == FACE_DIRICHLET;
        g.dir_hi[d]    = s.face[d][1] == FACE_DIRICHLET;
        const int h_lo = std::max(vertex ? 0 : prm.n_overlap - 1, s.kernel == ASM_KERNEL_TILED ? p : 0);
        const int h_hi = vertex ? (p - 1) : (prm.n_overlap - 1);
        s.halo_lo      = h_lo;
        s.halo_hi      = h_hi + 1;
        if (g.wrap[d])
          {
            g.ext_lo[d] = 0;
            g.ext_hi[d] = np - 1;
            g.own_hi[d] = np - 1;
          }
        else
          {
            g.ext_lo[d] = s.face[d][0] == FACE_INTERFACE ? -h_lo : 0;
            g.ext_hi[d] = s.face[d][1] == FACE_INTERFACE ? np + h_hi : np;
            g.own_hi[d] = s.face[d][1] == FACE_INTERFACE ? np - 1 : np;
          }
        s.store_size[d] = g.ext_hi[d] - g.ext_lo[d] + 1;
        s.n_owned *= g.own_hi[d] + 1;
        s.n_global *= (long long)s.ncg[d] * p + (periodic ? 0 : 1);
        s.ence[d] = g.nc[d] + 2;
      }
    const long long pitch = ((s.store_size[0] + 7) / 8) * 8;
    g.stride[0]           = 1;
    g.stride[1]           = pitch;
    g.stride[2]           = pitch * s.store_size[1];
    s.n_alloc             = g.stride[2] * s.store_size[2];
    s.n_ghost             = (long long)s.store_size[0] * s.store_size[1] * s.store_size[2] - s.n_owned;
    return true;
  }

  inline void
  fill_layout(const HostSetup &s, asm_layout *out)
  {
    out->n_global_dofs = s.n_global;
    out->n_owned       = s.n_owned;
    out->n_ghost       = s.n_ghost;
    out->n_alloc       = s.n_alloc;
    for (int d = 0; d < 3; ++d)
      {
        out->owned_begin[d] = s.cell_begin[d] * s.g.p;
        out->owned_size[d]  = s.g.own_hi[d] + 1;
        out->store_begin[d] = s.g.ext_lo[d];
        out->store_size[d]  = s.store_size[d];
        out->stride[d]      = s.g.stride[d];
        out->cell_begin[d]  = s.cell_begin[d];
        out->cell_size[d]   = s.g.nc[d];
        out->proc_grid[d]   = s.P[d];
        out->proc_coord[d]  = s.pc[d];
      }
  }

  // Single-round ghost update: one box to / from each of the up to 26 neighbouring blocks.
  // Per direction d a neighbour offset o_d selects a zone of planes:
  //   o_d =  0: the owned planes [0, own_hi] on both sides;
  //   o_d = +1: I send my top planes [np - n_lo, np) (its low ghosts), I receive its planes
  //             [0, n_hi) as my high ghosts [np, np + n_hi);
  //   o_d = -1: I send my planes [0, n_hi) (its high ghosts), I receive my low ghosts [-n_lo, 0).
  // (a shared plane belongs to the block on its high side).  The box of a message is the product
  // of its three zones; what I send towards o is what that neighbour receives from -o.
  // Order: sends in the order of the array (lexicographic in the offset), receives in reverse
  // order, so that between two ranks that are each other's neighbours more than once (two blocks in
  // a periodic direction) the k-th send meets the k-th receive.
  inline int
  build_direct_plan(const HostSetup &s, asm_direct_message msgs[26])
  {
    const DevGrid &g    = s.g;
    const int      n_lo = s.halo_lo, n_hi = s.halo_hi;
    int            n    = 0;
    long long      so = 0, ro = 0;
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox)
          {
            const int o[3] = {ox, oy, oz};
            if (ox == 0 && oy == 0 && oz == 0)
              continue;
            bool exists = true;
            int  pcn[3];
            for (int d = 0; d < 3; ++d)
              {
                pcn[d] = s.pc[d];
                if (o[d] == 0)
                  continue;
                if (s.face[d][o[d] > 0 ? 1 : 0] != FACE_INTERFACE)
                  exists = false;
                pcn[d] = imod(s.pc[d] + o[d], s.P[d]);
              }
            if (!exists)
              continue;
            asm_direct_message &m = msgs[n];
            m.peer                = pcn[0] + s.P[0] * (pcn[1] + s.P[1] * pcn[2]);
            long long sc = 1, rc = 1;
            for (int d = 0; d < 3; ++d)
              {
                const int np = g.nc[d] * g.p;
                m.offset[d]  = o[d];
                if (o[d] == 0)
                  {
                    m.send_lo[d] = m.recv_lo[d] = 0;
                    m.send_size[d] = m.recv_size[d] = g.own_hi[d] + 1;
                  }
                else if (o[d] > 0)
                  {
                    m.send_lo[d]   = np - n_lo;
                    m.send_size[d] = n_lo;
                    m.recv_lo[d]   = np;
                    m.recv_size[d] = n_hi;
                  }
                else
                  {
                    m.send_lo[d]   = 0;
                    m.send_size[d] = n_hi;
                    m.recv_lo[d]   = -n_lo;
                    m.recv_size[d] = n_lo;
                  }
                sc *= m.send_size[d];
                rc *= m.recv_size[d];
              }
            if (sc == 0 && rc == 0)
              continue;
            m.send_offset = so;
            m.recv_offset = ro;
            m.send_count  = sc;
            m.recv_count  = rc;
            so += sc;
            ro += rc;
            ++n;
          }
    return n;
  }

  // ---------------------------------------------------------------------
  // ghost exchange plan: dimension-by-dimension slabs (x, then y with the x ghosts,
  // then z), the structured-block equivalent of Partitioner::export_to_ghosted_array
  // (include/matrix_free_internal.h:13-47); add = true is the reverse
  // (import_from_ghosted_array with VectorOperation::add, :49-84) in the order z, y, x.
  // Boxes are in local logical DoF coordinates; index [0] = low neighbour, [1] = high.
  // ---------------------------------------------------------------------
  inline int
  build_exchange_plan(const HostSetup &s, const bool add, asm_exchange_step steps[3])
  {
    const DevGrid &g   = s.g;
    const int      n_lo = s.halo_lo, n_hi = s.halo_hi;
    int            n_steps = 0;
    for (int step = 0; step < 3; ++step)
      {
        const int d  = add ? 2 - step : step;
        const int np = g.nc[d] * g.p;
        if (g.wrap[d])
          continue;
        const bool lo_if = s.face[d][0] == FACE_INTERFACE, hi_if = s.face[d][1] == FACE_INTERFACE;
        if (!lo_if && !hi_if)
          continue;
        asm_exchange_step &st = steps[n_steps++];
        st.direction          = d;
        st.peer[0]            = lo_if ? s.nbr[d][0] : -1;
        st.peer[1]            = hi_if ? s.nbr[d][1] : -1;
        // tangential extents: lower directions carry their ghost planes (update: they were
        // filled before; add: they are forwarded afterwards), higher ones do not
        int lo[3], size[3];
        for (int e = 0; e < 3; ++e)
          {
            if (g.wrap[e])
              {
                lo[e]   = 0;
                size[e] = g.nc[e] * g.p;
              }
            else if (e < d)
              {
                lo[e]   = g.ext_lo[e];
                size[e] = g.ext_hi[e] - g.ext_lo[e] + 1;
              }
            else
              {
                lo[e]   = 0;
                size[e] = g.own_hi[e] + 1;
              }
          }
        // owned planes the neighbours ghost:   low neighbour: [0, n_hi)   high neighbour: [np - n_lo, np)
        // my ghost planes:                     low: [-n_lo, 0)            high: [np, np + n_hi)
        const int own_lo[2] = {0, np - n_lo}, own_n[2] = {lo_if ? n_hi : 0, hi_if ? n_lo : 0};
        const int gh_lo[2] = {-n_lo, np}, gh_n[2] = {lo_if ? n_lo : 0, hi_if ? n_hi : 0};
        for (int sd = 0; sd < 2; ++sd)
          {
            for (int e = 0; e < 3; ++e)
              {
                st.send_lo[sd][e] = st.recv_lo[sd][e] = lo[e];
                st.send_size[sd][e] = st.recv_size[sd][e] = size[e];
              }
            st.send_lo[sd][d]   = add ? gh_lo[sd] : own_lo[sd];
            st.send_size[sd][d] = add ? gh_n[sd] : own_n[sd];
            st.recv_lo[sd][d]   = add ? own_lo[sd] : gh_lo[sd];
            st.recv_size[sd][d] = add ? own_n[sd] : gh_n[sd];
          }
      }
    return n_steps;
  }

  // ---------------------------------------------------------------------
  // geometry
  // ---------------------------------------------------------------------
  struct Geometry
  {
    const HostSetup &s;
    explicit Geometry(const HostSetup &s)
      : s(s)
    {}

    // global cell coordinate of local cell coordinate c (may be -1 or nc), or -1
    int global_cell(const int d, const int c) const
    {
      int cg = s.cell_begin[d] + c;
      if (s.prm.boundary[d][0] == ASM_BC_PERIODIC)
        return imod(cg, s.ncg[d]);
      return (cg >= 0 && cg < s.ncg[d]) ? cg : -1;
    }

    // 27 physical support points of the degree-2 mapping of global cell cg
    // (MappingQCache::initialize with the transformation function,
    //  element_centered_preconditioners_01.cc:419-424), index 9k+3j+i
    void q2_points(const int cg[3], double pts[27][3]) const
    {
      if (s.prm.geometry == ASM_GEOM_Q2_POINTS)
        {
          const long long cell = cg[0] + (long long)s.ncg[0] * (cg[1] + (long long)s.ncg[1] * cg[2]);
          std::memcpy(pts, s.prm.q2_points + cell * 81, 81 * sizeof(double));
          return;
        }
      for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j)
          for (int i = 0; i < 3; ++i)
            {
              const double xh = (cg[0] + 0.5 * i) / s.ncg[0], yh = (cg[1] + 0.5 * j) / s.ncg[1],
                           zh = (cg[2] + 0.5 * k) / s.ncg[2];
              double X[3]     = {xh, yh, zh};
              if (s.prm.geometry == ASM_GEOM_KERSHAW)
                kershaw_map(s.prm.kershaw_eps[0], s.prm.kershaw_eps[1], xh, yh, zh, X);
              for (int e = 0; e < 3; ++e)
                pts[9 * k + 3 * j + i][e] = X[e] * s.prm.extent[e];
            }
    }

    // nodal -> monomial coefficients direction by direction, operator.h:625-658
    static void to_monomial(double v[27][3])
    {
      static const double C[3][3] = {{1., 0., 0.}, {-3., 4., -1.}, {2., -4., 2.}};
      const int           str[3]  = {1, 3, 9};
      for (int d = 0; d < 3; ++d)
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b)
            {
              const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
              const int base = a * str[d1] + b * str[d2];
              for (int e = 0; e < 3; ++e)
                {
                  const double x0 = v[base][e], x1 = v[base + str[d]][e], x2 = v[base + 2 * str[d]][e];
                  for (int r = 0; r < 3; ++r)
                    v[base + r * str[d]][e] = C[r][0] * x0 + C[r][1] * x1 + C[r][2] * x2;
                }
            }
    }

    static void eval_monomial(const double v[27][3], const double xi[3], double X[3])
    {
      X[0] = X[1] = X[2] = 0;
      double pw[3][3];
      for (int d = 0; d < 3; ++d)
        {
          pw[d][0] = 1;
          pw[d][1] = xi[d];
          pw[d][2] = xi[d] * xi[d];
        }
      for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j)
          for (int i = 0; i < 3; ++i)
            {
              const double w = pw[0][i] * pw[1][j] * pw[2][k];
              for (int e = 0; e < 3; ++e)
                X[e] += w * v[9 * k + 3 * j + i][e];
            }
    }

    // GridTools::compute_harmonic_cell_extend, include/grid_tools.h:11-50
    void harmonic_extent(const double v[27][3], const Basis1D &b, double ext[3]) const
    {
      for (int d = 0; d < 3; ++d)
        {
          const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
          double    sum = 0;
          for (int q2 = 0; q2 < b.n1; ++q2)
            for (int q1 = 0; q1 < b.n1; ++q1)
              {
                double xi[3];
                xi[d1] = b.xq[q1];
                xi[d2] = b.xq[q2];
                double X0[3], X1[3];
                xi[d] = 0;
                eval_monomial(v, xi, X0);
                xi[d] = 1;
                eval_monomial(v, xi, X1);
                const double dist = std::sqrt((X1[0] - X0[0]) * (X1[0] - X0[0]) + (X1[1] - X0[1]) * (X1[1] - X0[1]) +
                                              (X1[2] - X0[2]) * (X1[2] - X0[2]));
                sum += dist * b.wq[q1] * b.wq[q2];
              }
          ext[d] = sum;
        }
    }
  };

  // ---------------------------------------------------------------------
  // 1-D patch matrices
  // ---------------------------------------------------------------------
  // element-centred: [deal.II upstream, call site include/matrix_free.h:350-360]
  // bc: 0 = interior face (neighbour exists), ASM_BC_DIRICHLET, ASM_BC_NEUMANN
  inline void
  fdm_matrices_element(const Basis1D &b, const double h3[3], const int bc_lo, const int bc_hi, const int n_overlap,
                       std::vector<double> &M, std::vector<double> &K)
  {
    const int n1 = b.n1, m = n1 - 2 + 2 * n_overlap, o = n_overlap - 1;
    M.assign(m * m, 0.);
    K.assign(m * m, 0.);
    for (int i = 0; i < n1; ++i)
      for (int j = 0; j < n1; ++j)
        {
          M[(i + o) * m + j + o] = b.M_ref[i * n1 + j] * h3[1];
          K[(i + o) * m + j + o] = b.K_ref[i * n1 + j] / h3[1];
        }
    const auto clear = [&](const int r) {
      for (int j = 0; j < m; ++j)
        M[r * m + j] = M[j * m + r] = K[r * m + j] = K[j * m + r] = 0.;
    };
    if (bc_lo == 0)
      {
        for (int i = 0; i < n_overlap; ++i)
          for (int j = 0; j < n_overlap; ++j)
            {
              M[i * m + j] += b.M_ref[(n1 - n_overlap + i) * n1 + n1 - n_overlap + j] * h3[0];
              K[i * m + j] += b.K_ref[(n1 - n_overlap + i) * n1 + n1 - n_overlap + j] / h3[0];
            }
      }
    else if (bc_lo == ASM_BC_DIRICHLET)
      clear(o);
    if (bc_hi == 0)
      {
        const int sft = n_overlap + n1 - 2;
        for (int i = 0; i < n_overlap; ++i)
          for (int j = 0; j < n_overlap; ++j)
            {
              M[(sft + i) * m + sft + j] += b.M_ref[i * n1 + j] * h3[2];
              K[(sft + i) * m + sft + j] += b.K_ref[i * n1 + j] / h3[2];
            }
      }
    else if (bc_hi == ASM_BC_DIRICHLET)
      clear(n_overlap + n1 - 2);
  }

  // vertex-star: include/tensor_product_matrix_creator.h:7-61
  inline void
  fdm_matrices_vertex(const Basis1D &b, const double h2[2], std::vector<double> &M, std::vector<double> &K)
  {
    const int n1 = b.n1, m = 2 * (n1 - 1) - 1;
    M.assign(m * m, 0.);
    K.assign(m * m, 0.);
    for (int i = 0; i < n1 - 1; ++i)
      for (int j = 0; j < n1 - 1; ++j)
        {
          M[i * m + j] += b.M_ref[(i + 1) * n1 + j + 1] * h2[0];
          K[i * m + j] += b.K_ref[(i + 1) * n1 + j + 1] / h2[0];
        }
    for (int i = 0; i < n1 - 1; ++i)
      for (int j = 0; j < n1 - 1; ++j)
        {
          M[(i + n1 - 2) * m + j + n1 - 2] += b.M_ref[i * n1 + j] * h2[1];
          K[(i + n1 - 2) * m + j + n1 - 2] += b.K_ref[i * n1 + j] / h2[1];
        }
  }

  // fn(begin, end) on contiguous slices of [0, n) with the host's threads (ASM_SETUP_THREADS, at most 32)
  template <typename F>
  inline void
  parallel_for(const long long n, const F &fn)
  {
    int nt = (int)std::min<long long>(std::max(1u, std::thread::hardware_concurrency()), 32);
    if (const char *e = std::getenv("ASM_SETUP_THREADS"))
      nt = std::max(1, std::atoi(e));
    nt = (int)std::min<long long>(nt, std::max<long long>(n / 64, 1));
    if (nt <= 1)
      {
        fn(0, n);
        return;
      }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th)
      x.join();
  }

  // ---------------------------------------------------------------------
  // full setup
  // ---------------------------------------------------------------------
  inline bool
  setup_all(HostSetup &s, const Basis1D &b)
  {
    const asm_params &prm    = s.prm;
    const DevGrid &   g      = s.g;
    const int         p      = g.p;
    const bool        vertex = prm.element_centric == 0;
    const int         n      = prm.n_overlap;
    const int         m      = s.m;
    Geometry          geo(s);
    s.cartesian = prm.geometry == ASM_GEOM_CARTESIAN;
    if (prm.geometry == ASM_GEOM_Q2_POINTS && prm.q2_points == nullptr)
      {
        s.error = "ASM_GEOM_Q2_POINTS needs q2_points";
        return false;
      }
    if (prm.geometry != ASM_GEOM_CARTESIAN)
      for (int d = 0; d < 3; ++d)
        if (prm.boundary[d][0] == ASM_BC_PERIODIC)
          {
            s.error = "periodic boundaries are supported on Cartesian geometry only";
            return false;
          }

    const long long n_ext = (long long)s.ence[0] * s.ence[1] * s.ence[2];
    s.cell_exists.assign(n_ext, 0);

    // ---- per-cell geometry on the extended cell box + harmonic extents
    std::vector<double> ext; // [ext cells][3]
    if (!s.cartesian)
      {
        s.coeff.assign(n_ext * 81, 0.);
        ext.assign(n_ext * 3, 0.);
      }
    for (int d = 0; d < 3; ++d)
      {
        s.h[d].assign(s.ence[d], 0.);
        for (int c = -1; c <= g.nc[d]; ++c)
          if (geo.global_cell(d, c) >= 0)
            s.h[d][c + 1] = prm.extent[d] / s.ncg[d];
      }
    // (planes of cells are independent: threaded, the deformed case evaluates the geometry of every cell)
    parallel_for(g.nc[2] + 2, [&](const long long z0, const long long z1) {
      for (int cz = (int)z0 - 1; cz < (int)z1 - 1; ++cz)
        for (int cy = -1; cy <= g.nc[1]; ++cy)
          for (int cx = -1; cx <= g.nc[0]; ++cx)
            {
              const int cg[3] = {geo.global_cell(0, cx), geo.global_cell(1, cy), geo.global_cell(2, cz)};
              const int ce    = s.ext_cell(cx, cy, cz);
              if (cg[0] < 0 || cg[1] < 0 || cg[2] < 0)
                continue;
              s.cell_exists[ce] = 1;
              if (!s.cartesian)
                {
                  double v[27][3];
                  geo.q2_points(cg, v);
                  Geometry::to_monomial(v);
                  std::memcpy(&s.coeff[(long long)ce * 81], v, 81 * sizeof(double));
                  geo.harmonic_extent(v, b, &ext[(long long)ce * 3]);
                }
            }
    });

    if (prm.precond_type != ASM_PRECOND_FDM)
      {
        s.n_inst = 0;
        s.has_pre = s.has_post = false;
        return true;
      }

    // ---- FDM instances ---------------------------------------------------
    std::map<std::vector<double>, int> cache;
    s.eig.clear();
    s.eig_sym.clear();
    const auto get_instance = [&](const double h3[3], const int bc_lo, const int bc_hi) -> int {
      std::vector<double> key = {h3[0], h3[1], h3[2], (double)bc_lo, (double)bc_hi};
      if (vertex)
        key = {h3[0], h3[1]};
      const auto it = cache.find(key);
      if (it != cache.end())
        return it->second;
      std::vector<double> M, K, lam, Q;
      if (vertex)
        fdm_matrices_vertex(b, h3, M, K);
      else
        fdm_matrices_element(b, h3, bc_lo, bc_hi, n, M, K);
      generalized_eigen(m, M, K, lam, Q);
      s.eig_sym.push_back(sort_eigenvectors_by_parity(m, Q, lam) ? 1 : 0);
      const int id = cache.size();
      cache[key]   = id;
      s.eig.insert(s.eig.end(), Q.begin(), Q.end());
      s.eig.insert(s.eig.end(), lam.begin(), lam.end());
      return id;
    };
    // boundary type seen by a cell on its low/high face in direction d
    const auto face_bc = [&](const int d, const int c, const int side) -> int {
      const int cn = geo.global_cell(d, c + (side ? 1 : -1));
      if (cn >= 0)
        return 0;
      return prm.boundary[d][side];
    };
    s.tensor_inst = s.cartesian;
    if (s.tensor_inst)
      {
        for (int d = 0; d < 3; ++d)
          {
            s.inst[d].assign(s.ence[d], 0);
            for (int c = -1; c <= g.nc[d]; ++c)
              {
                if (geo.global_cell(d, c) < 0)
                  continue;
                const double hh = prm.extent[d] / s.ncg[d];
                const int    lo = face_bc(d, c, 0), hi = face_bc(d, c, 1);
                if (vertex)
                  {
                    // collect_patch_extend, matrix_free.h:1510-1524
                    const double h2[3] = {hh, hi == 0 ? hh : 1.0, 0.};
                    s.inst[d][c + 1]   = get_instance(h2, 0, 0);
                  }
                else
                  {
                    const double h3[3] = {lo == 0 ? hh : 0., hh, hi == 0 ? hh : 0.};
                    s.inst[d][c + 1]   = get_instance(h3, lo, hi);
                  }
              }
          }
      }
    else
      {
        for (int d = 0; d < 3; ++d)
          s.inst[d].assign(n_ext, 0);
        // One instance per cell and direction, nearly all of them different on a deformed mesh: (1) the keys (extents of
        // the cell and its two neighbours, boundary kinds) of all cells, threaded; (2) numbering of the distinct keys in
        // cell order; (3) one generalized eigenproblem per distinct key, threaded.
        std::vector<double> keys((size_t)n_ext * 3 * 5, 0.);
        parallel_for(g.nc[2] + 2, [&](const long long z0, const long long z1) {
          for (int cz = (int)z0 - 1; cz < (int)z1 - 1; ++cz)
            for (int cy = -1; cy <= g.nc[1]; ++cy)
              for (int cx = -1; cx <= g.nc[0]; ++cx)
                {
                  const int ce = s.ext_cell(cx, cy, cz);
                  if (!s.cell_exists[ce])
                    continue;
                  const int c3[3] = {cx, cy, cz};
                  for (int d = 0; d < 3; ++d)
                    {
                      // neighbours' extents: only available inside the extended box;
                      // cells on the rim of the extended box are halo cells whose
                      // outer neighbours are never needed by owned DoFs of n_overlap=1
                      // patches, but their own instance must still be right, so
                      // evaluate the neighbour geometry directly.
                      double h3[3] = {0., ext[(long long)ce * 3 + d], 0.};
                      for (int side = 0; side < 2; ++side)
                        {
                          int cn[3] = {c3[0], c3[1], c3[2]};
                          cn[d] += side ? 1 : -1;
                          const int cg[3] = {geo.global_cell(0, cn[0]), geo.global_cell(1, cn[1]), geo.global_cell(2, cn[2])};
                          if (cg[d] < 0)
                            continue;
                          double e3[3];
                          if (cn[d] >= -1 && cn[d] <= g.nc[d])
                            {
                              const int cne = s.ext_cell(cn[0], cn[1], cn[2]);
                              e3[d]         = ext[(long long)cne * 3 + d];
                            }
                          else
                            {
                              double v[27][3];
                              geo.q2_points(cg, v);
                              Geometry::to_monomial(v);
                              geo.harmonic_extent(v, b, e3);
                            }
                          h3[side ? 2 : 0] = e3[d];
                        }
                      double *key = &keys[((size_t)ce * 3 + d) * 5];
                      if (vertex)
                        {
                          key[0] = h3[1] != 0. ? h3[1] : 1.0;
                          key[1] = h3[2] != 0. ? h3[2] : 1.0;
                        }
                      else
                        {
                          key[0] = h3[0];
                          key[1] = h3[1];
                          key[2] = h3[2];
                          key[3] = face_bc(d, c3[d], 0);
                          key[4] = face_bc(d, c3[d], 1);
                        }
                    }
                }
        });
        std::map<std::array<double, 5>, int> ids;
        std::vector<std::array<double, 5>>   distinct;
        for (int cz = -1; cz <= g.nc[2]; ++cz)
          for (int cy = -1; cy <= g.nc[1]; ++cy)
            for (int cx = -1; cx <= g.nc[0]; ++cx)
              {
                const int ce = s.ext_cell(cx, cy, cz);
                if (!s.cell_exists[ce])
                  continue;
                for (int d = 0; d < 3; ++d)
                  {
                    std::array<double, 5> key;
                    std::memcpy(key.data(), &keys[((size_t)ce * 3 + d) * 5], 5 * sizeof(double));
                    const auto it = ids.emplace(key, (int)distinct.size());
                    if (it.second)
                      distinct.push_back(key);
                    s.inst[d][ce] = it.first->second;
                  }
              }
        const int E = m * m + m;
        s.eig.assign(distinct.size() * (size_t)E, 0.);
        s.eig_sym.assign(distinct.size(), 0);
        parallel_for((long long)distinct.size(), [&](const long long i0, const long long i1) {
          std::vector<double> M, K, lam, Q;
          for (long long id = i0; id < i1; ++id)
            {
              const std::array<double, 5> &key = distinct[id];
              const double                 h3[3] = {key[0], key[1], key[2]};
              if (vertex)
                fdm_matrices_vertex(b, h3, M, K);
              else
                fdm_matrices_element(b, h3, (int)key[3], (int)key[4], n, M, K);
              generalized_eigen(m, M, K, lam, Q);
              s.eig_sym[id] = sort_eigenvectors_by_parity(m, Q, lam) ? 1 : 0;
              std::copy(Q.begin(), Q.end(), s.eig.begin() + id * E);
              std::copy(lam.begin(), lam.end(), s.eig.begin() + id * E + m * m);
            }
        });
        s.n_inst = (int)distinct.size();
      }
    if (s.tensor_inst)
      s.n_inst = cache.size();


    // ---- weights: tensor products of 1-D tables ---------------------------
    // valence (include/matrix_free.h:676-712) and RAS ownership (:536-673) are
    // products over the directions on a structured mesh.
    const int wt = prm.weighting_type;
    s.has_pre  = (wt == ASM_W_PRE || wt == ASM_W_SYMM);
    s.has_post = (wt == ASM_W_POST || wt == ASM_W_SYMM || wt == ASM_W_RAS);
    for (int d = 0; d < 3; ++d)
      {
        const bool periodic = prm.boundary[d][0] == ASM_BC_PERIODIC;
        const int  Ng       = s.ncg[d] * p; // last global plane (non-periodic) / period
        // patch of global cell cg: origin and validity of global coordinate Ig
        const auto patch_origin = [&](const int cg) { return vertex ? p * cg + 1 : p * cg - (n - 1); };
        const auto patch_ok     = [&](const int cg) {
          if (cg < 0 || cg >= s.ncg[d])
            return false;
          if (vertex && !periodic && cg + 1 >= s.ncg[d])
            return false;
          return true;
        };
        const auto coord_valid = [&](const int Ig) {
          if (periodic)
            return true;
          if (Ig < 0 || Ig > Ng)
            return false;
          if (Ig == 0 && prm.boundary[d][0] == ASM_BC_DIRICHLET)
            return false;
          if (Ig == Ng && prm.boundary[d][1] == ASM_BC_DIRICHLET)
            return false;
          return true;
        };
        // cells (unwrapped coordinate) whose patch contains unwrapped Ig
        const auto valence = [&](const int Ig, int &owner) {
          int v = 0;
          owner = 1 << 30;
          for (int cu = (Ig - 3 * m) / p - 2; cu <= (Ig + 3 * m) / p + 2; ++cu)
            {
              const int cg = periodic ? imod(cu, s.ncg[d]) : cu;
              if (!patch_ok(cg))
                continue;
              const int o = patch_origin(cu);
              if (Ig < o || Ig >= o + m)
                continue;
              ++v;
              // RAS core region (matrix_free.h:575-582)
              const int  i       = Ig - o;
              const bool in_core = vertex ? true : (i >= n - 1 && i < p + n);
              if (in_core)
                owner = std::min(owner, cg);
            }
          return v;
        };
        s.w_pre[d].assign((size_t)s.ence[d] * m, 0.);
        s.w_post[d].assign((size_t)s.ence[d] * m, 0.);
        for (int c = -1; c <= g.nc[d]; ++c)
          {
            const int cu = s.cell_begin[d] + c; // unwrapped global cell coordinate
            const int cg = periodic ? imod(cu, s.ncg[d]) : cu;
            if (!patch_ok(cg))
              continue;
            for (int i = 0; i < m; ++i)
              {
                const int Ig = patch_origin(cu) + i;
                if (!coord_valid(Ig))
                  continue;
                int          owner;
                const int    v   = valence(Ig, owner);
                const double val = v;
                double       wpre = 1., wpost = 1.;
                if (wt == ASM_W_PRE)
                  wpre = 1. / val;
                else if (wt == ASM_W_POST)
                  wpost = 1. / val;
                else if (wt == ASM_W_SYMM)
                  wpre = wpost = 1. / std::sqrt(val);
                else if (wt == ASM_W_RAS)
                  wpost = (owner == cg) ? 1. : 0.;
                s.w_pre[d][(size_t)(c + 1) * m + i]  = wpre;
                s.w_post[d][(size_t)(c + 1) * m + i] = wpost;
              }
          }
      }
    return true;
  }
} // namespace asmamd
