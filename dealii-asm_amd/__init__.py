"""ctypes binding of libdealii_asm_amd.so (the C ABI in include/dealii_asm_amd.h).

Host-side mirror, for tests and bench.py, of the reference's operator surface:
  LaplaceOperatorMatrixFree.vmult / compute_inverse_diagonal  (include/operator.h)
  ASPoissonPreconditioner.vmult                               (include/matrix_free.h)
  PreconditionChebyshev.vmult / step / estimate_eigenvalues   (deal.II, configured in
                                                               include/precondition.templates.h:89-158)
The product path has NO CPU fallback: importing works without a GPU (so that the
exported symbols can be checked), creating a Context without one raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEALII_ASM_AMD_LIB", os.path.join(_HERE, "libdealii_asm_amd.so"))

BC = {"periodic": 0, "dirichlet": 1, "neumann": 2}
WEIGHTING = {"none": 0, "pre": 1, "post": 2, "ras": 3, "symm": 4}
GEOMETRY = {"cartesian": 0, "kershaw": 1, "q2_points": 2}
PRECOND = {"FDM": 0, "Diagonal": 1}
EV = {"auto": 0, "lanczos": 1, "power iteration": 2}
KERNEL = {"auto": 0, "generic": 1, "tiled": 2}
POLYNOMIAL = {"1st kind": 0, "4th kind": 1}
FUSED = {"none": 0, "residual": 1, "cheb_first": 2, "relax": 3, "cheb_next": 4}
MG_TYPE = {"h": 0, "p": 1, "hp": 2, "ph": 3}
MG_P_SEQUENCE = {"bisect": 0, "decrease by one": 1, "go to one": 2}
MG_COARSE = {"smoother": 0, "cg": 1, "direct": 2, "auto": 3}
SOLVER = {"CG": 0, "GMRES": 1}
SYSTEM_PRECONDITIONER = {"Identity": 0, "inner": 1, "Chebyshev": 2, "Multigrid": 3}


class AsmParams(C.Structure):
    _fields_ = [
        ("degree", C.c_int32),
        ("n_cells", C.c_int32 * 3),
        ("boundary", (C.c_int32 * 2) * 3),
        ("geometry", C.c_int32),
        ("extent", C.c_double * 3),
        ("kershaw_eps", C.c_double * 2),
        ("q2_points", C.POINTER(C.c_double)),
        ("precond_type", C.c_int32),
        ("n_overlap", C.c_int32),
        ("weighting_type", C.c_int32),
        ("element_centric", C.c_int32),
        ("cheb_degree", C.c_int32),
        ("smoothing_range", C.c_double),
        ("eig_n_iterations", C.c_int32),
        ("ev_algorithm", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("rank", C.c_int32),
        ("world_size", C.c_int32),
        ("proc_grid", C.c_int32 * 3),
        ("polynomial_type", C.c_int32),
        ("struct_size", C.c_uint32),
    ]


class AsmLayout(C.Structure):
    _fields_ = [
        ("n_global_dofs", C.c_int64),
        ("n_owned", C.c_int64),
        ("n_ghost", C.c_int64),
        ("n_alloc", C.c_int64),
        ("owned_begin", C.c_int32 * 3),
        ("owned_size", C.c_int32 * 3),
        ("store_begin", C.c_int32 * 3),
        ("store_size", C.c_int32 * 3),
        ("stride", C.c_int64 * 3),
        ("cell_begin", C.c_int32 * 3),
        ("cell_size", C.c_int32 * 3),
        ("proc_grid", C.c_int32 * 3),
        ("proc_coord", C.c_int32 * 3),
    ]


class AsmDirectMessage(C.Structure):
    _fields_ = [
        ("peer", C.c_int32),
        ("offset", C.c_int32 * 3),
        ("send_lo", C.c_int32 * 3),
        ("send_size", C.c_int32 * 3),
        ("recv_lo", C.c_int32 * 3),
        ("recv_size", C.c_int32 * 3),
        ("send_offset", C.c_int64),
        ("recv_offset", C.c_int64),
        ("send_count", C.c_int64),
        ("recv_count", C.c_int64),
    ]


class AsmExchangeStep(C.Structure):
    _fields_ = [
        ("direction", C.c_int32),
        ("peer", C.c_int32 * 2),
        ("send_lo", (C.c_int32 * 3) * 2),
        ("send_size", (C.c_int32 * 3) * 2),
        ("recv_lo", (C.c_int32 * 3) * 2),
        ("recv_size", (C.c_int32 * 3) * 2),
    ]


class AsmMgParams(C.Structure):
    _fields_ = [
        ("mg_type", C.c_int32),
        ("p_sequence", C.c_int32),
        ("one_sided_v_cycle", C.c_int32),
        ("coarse_solver", C.c_int32),
        ("coarse_rel_tolerance", C.c_double),
        ("coarse_max_iterations", C.c_int32),
        ("estimate_eigenvalues", C.c_int32),
        ("n_coarse_cycles", C.c_int32),
        ("intermediate_smoother", C.c_int32),
        ("intermediate_precond_type", C.c_int32),
        ("intermediate_cheb_degree", C.c_int32),
        ("keep_duplicate_level", C.c_int32),
        ("struct_size", C.c_uint32),
    ]


class AsmSolverParams(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("max_iterations", C.c_int32),
        ("abs_tolerance", C.c_double),
        ("rel_tolerance", C.c_double),
        ("right_preconditioning", C.c_int32),
        ("max_n_tmp_vectors", C.c_int32),
        ("keep_preconditioned_basis", C.c_int32),
        ("struct_size", C.c_uint32),
    ]


class AsmSolverResult(C.Structure):
    _fields_ = [
        ("n_iterations", C.c_int32),
        ("converged", C.c_int32),
        ("initial_residual", C.c_double),
        ("final_residual", C.c_double),
    ]


# every symbol declared in include/dealii_asm_amd.h
EXPORTED_SYMBOLS = [
    "asm_params_default",
    "asm_params_set",
    "asm_create",
    "asm_destroy",
    "asm_last_error",
    "asm_describe_layout",
    "asm_exchange_plan",
    "asm_exchange_plan_direct",
    "asm_comm_unique_id",
    "asm_comm_init",
    "asm_vector_layout",
    "asm_vector_alloc",
    "asm_vector_free",
    "asm_vector_import",
    "asm_vector_export",
    "asm_update_ghost_values",
    "asm_compress_add",
    "asm_op_vmult",
    "asm_op_inverse_diagonal",
    "asm_precond_vmult",
    "asm_op_vmult_fused",
    "asm_precond_vmult_fused",
    "asm_is_symmetric",
    "asm_cheb_estimate_eigenvalues",
    "asm_cheb_set_eigenvalue",
    "asm_cheb_vmult",
    "asm_cheb_step",
    "asm_n_fdm_instances",
    "asm_memory_consumption",
    "asm_cheb_parameters",
    "asm_kernel_variant_in_use",
    "asm_synchronize",
    "asm_time_calls",
    "asm_set_keep_ghosts",
    "asm_debug_external_exchange",
    "asm_debug_two_phase",
    "asm_debug_loopback_init",
    "asm_mg_debug_after_loopback",
    "asm_debug_exchange_all",
    "asm_profile_enable",
    "asm_profile_get",
    "asm_measure_copy_bandwidth",
    "asm_tiled_laplace_rows",
    "asm_vec_dot",
    "asm_vec_multi_dot",
    "asm_vec_multi_axpy_norm",
    "asm_vec_axpby",
    "asm_vec_scale",
    "asm_vec_copy",
    "asm_vec_zero",
    "asm_vec_constraints_set_zero",
    "asm_mg_params_default",
    "asm_mg_create",
    "asm_mg_destroy",
    "asm_mg_last_error",
    "asm_mg_n_levels",
    "asm_mg_level_context",
    "asm_mg_prolongate_add",
    "asm_mg_restrict",
    "asm_mg_vmult",
    "asm_tiled_plan",
    "asm_pair_plan",
    "asm_pair_sweeps_in_use",
    "asm_pair_tiles",
    "asm_pair_zchunks",
    "asm_pair_waves",
    "asm_mg_coarse_failures",
    "asm_mg_graph_replays",
    "asm_mg_comm_init",
    "asm_mg_level_distributed",
    "asm_solver_params_default",
    "asm_solve",
]

_lib = None


def load_library():
    """dlopen the in-tree library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C dealii-asm_amd` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    lib.asm_last_error.restype = C.c_char_p
    lib.asm_last_error.argtypes = [C.c_void_p]
    lib.asm_params_default.argtypes = [C.POINTER(AsmParams)]
    lib.asm_params_default.restype = None
    lib.asm_params_set.argtypes = [C.POINTER(AsmParams), C.c_char_p, C.c_char_p]
    lib.asm_op_vmult_fused.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.asm_precond_vmult_fused.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    lib.asm_is_symmetric.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_create.argtypes = [C.POINTER(AsmParams), C.POINTER(C.c_void_p)]
    lib.asm_destroy.argtypes = [C.c_void_p]
    lib.asm_destroy.restype = None
    lib.asm_describe_layout.argtypes = [C.POINTER(AsmParams), C.POINTER(AsmLayout)]
    lib.asm_exchange_plan.argtypes = [C.POINTER(AsmParams), C.c_int, C.POINTER(AsmExchangeStep), C.POINTER(C.c_int32)]
    lib.asm_exchange_plan_direct.argtypes = [C.POINTER(AsmParams), C.POINTER(AsmDirectMessage), C.POINTER(C.c_int32)]
    lib.asm_vector_layout.argtypes = [C.c_void_p, C.POINTER(AsmLayout)]
    lib.asm_vector_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.asm_vector_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.asm_vector_import.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.asm_vector_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("asm_update_ghost_values", "asm_compress_add", "asm_op_inverse_diagonal"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
    for name in ("asm_op_vmult", "asm_precond_vmult", "asm_cheb_vmult", "asm_cheb_step"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.asm_cheb_estimate_eigenvalues.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.asm_cheb_set_eigenvalue.argtypes = [C.c_void_p, C.c_double]
    lib.asm_n_fdm_instances.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.asm_memory_consumption.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.asm_cheb_parameters.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.asm_kernel_variant_in_use.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_synchronize.argtypes = [C.c_void_p]
    lib.asm_time_calls.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.asm_set_keep_ghosts.argtypes = [C.c_void_p, C.c_int]
    lib.asm_debug_external_exchange.argtypes = [C.c_void_p, C.c_int]
    lib.asm_debug_two_phase.argtypes = [C.c_void_p, C.c_int]
    lib.asm_debug_loopback_init.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.asm_debug_exchange_all.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int]
    lib.asm_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.asm_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.asm_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    lib.asm_tiled_laplace_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_pair_sweeps_in_use.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.asm_pair_tiles.argtypes = [C.POINTER(AsmParams), C.c_int32, C.POINTER(C.c_int32 * 6), C.c_int32, C.POINTER(C.c_int32)]
    lib.asm_pair_zchunks.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_pair_waves.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_pair_plan.argtypes = [C.POINTER(AsmParams), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
    lib.asm_vec_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.asm_vec_multi_dot.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_double)]
    lib.asm_vec_multi_axpy_norm.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.c_int32,
                                            C.POINTER(C.c_double)]
    lib.asm_vec_axpby.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double]
    lib.asm_vec_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.asm_vec_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.asm_vec_zero.argtypes = [C.c_void_p, C.c_void_p]
    lib.asm_vec_constraints_set_zero.argtypes = [C.c_void_p, C.c_void_p]
    lib.asm_mg_params_default.argtypes = [C.POINTER(AsmMgParams)]
    lib.asm_mg_params_default.restype = None
    lib.asm_mg_create.argtypes = [C.POINTER(AsmParams), C.POINTER(AsmMgParams), C.POINTER(C.c_void_p)]
    lib.asm_mg_destroy.argtypes = [C.c_void_p]
    lib.asm_mg_destroy.restype = None
    lib.asm_mg_last_error.argtypes = [C.c_void_p]
    lib.asm_mg_last_error.restype = C.c_char_p
    lib.asm_mg_n_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.asm_mg_level_context.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.asm_mg_prolongate_add.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.asm_mg_restrict.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.asm_mg_vmult.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.asm_mg_comm_init.argtypes = [C.c_void_p, C.c_char_p]
    lib.asm_mg_level_distributed.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.asm_mg_debug_after_loopback.argtypes = [C.c_void_p]
    lib.asm_mg_coarse_failures.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.asm_tiled_plan.argtypes = [C.POINTER(AsmParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                   C.POINTER(C.c_int32)]
    lib.asm_mg_graph_replays.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.asm_solver_params_default.argtypes = [C.POINTER(AsmSolverParams)]
    lib.asm_solver_params_default.restype = None
    lib.asm_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(AsmSolverParams), C.c_void_p, C.c_void_p, C.POINTER(AsmSolverResult)]
    lib.asm_comm_unique_id.argtypes = [C.c_char_p]
    lib.asm_comm_init.argtypes = [C.c_void_p, C.c_char_p]
    _lib = lib
    return lib


def tiled_plan(params, n_cu=256, rows=0):
    """asm_tiled_plan: (plan dict, order list) of the degree-4 tiled sweeps for the block `params` describes (host only)"""
    lib = load_library()
    plan, n = (C.c_int32 * 8)(), C.c_int32()
    cap = 1 << 16
    order = (C.c_int32 * cap)()
    err = lib.asm_tiled_plan(C.byref(params), n_cu, rows, plan, order, cap, C.byref(n))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    keys = ("ntx", "nty", "nzc", "zlen", "x_first", "y_first", "owned_x", "owned_y")
    return dict(zip(keys, list(plan))), list(order[: n.value])


def pair_plan(params, n_cu=256):
    """asm_pair_plan: (plan dict, order list) of the degree-4 wave-pair sweeps for the block `params` describes (host only)"""
    lib = load_library()
    plan, n = (C.c_int32 * 6)(), C.c_int32()
    cap = 1 << 16
    order = (C.c_int32 * cap)()
    err = lib.asm_pair_plan(C.byref(params), n_cu, plan, order, cap, C.byref(n))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    keys = ("ntx", "nty", "nzc", "zlen", "x_first", "y_first")
    return dict(zip(keys, list(plan))), list(order[: n.value])


def pair_tiles(params, n_cu=256):
    """asm_pair_tiles: [(x0, y0, cells x, cells y, owns first column | owns first row << 1, last ownable column), ...]"""
    lib = load_library()
    n = C.c_int32()
    cap = 1 << 16
    tiles = ((C.c_int32 * 6) * cap)()
    err = lib.asm_pair_tiles(C.byref(params), n_cu, tiles, cap, C.byref(n))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    return [tuple(tiles[i]) for i in range(n.value)]


def make_params(
    degree=4,
    n_cells=(8, 8, 8),
    boundary="dirichlet",
    geometry="cartesian",
    extent=(1.0, 1.0, 1.0),
    kershaw_eps=(1.0, 1.0),
    q2_points=None,
    precond="FDM",
    n_overlap=1,
    weighting="post",
    element_centric=True,
    cheb_degree=3,
    smoothing_range=20.0,
    eig_n_iterations=40,
    ev_algorithm="auto",
    kernel="auto",
    device=-1,
    stream=None,
    rank=0,
    world_size=1,
    proc_grid=(0, 0, 0),
    polynomial_type="1st kind",
    options=None,
):
    """Fill an asm_params with the reference's key names as keyword arguments
    (include/precondition.templates.h:108-152, 195-211)."""
    lib = load_library()
    p = AsmParams()
    lib.asm_params_default(C.byref(p))
    p.degree = degree
    if isinstance(boundary, str):
        boundary = [(boundary, boundary)] * 3
    for d in range(3):
        p.n_cells[d] = int(n_cells[d])
        b = boundary[d]
        if isinstance(b, str):
            b = (b, b)
        p.boundary[d][0] = BC[b[0]]
        p.boundary[d][1] = BC[b[1]]
        p.extent[d] = float(extent[d])
        p.proc_grid[d] = int(proc_grid[d])
    p.geometry = GEOMETRY[geometry]
    p.kershaw_eps[0], p.kershaw_eps[1] = float(kershaw_eps[0]), float(kershaw_eps[1])
    keep = None
    if q2_points is not None:
        keep = np.ascontiguousarray(q2_points, dtype=np.float64)
        p.q2_points = keep.ctypes.data_as(C.POINTER(C.c_double))
    p.precond_type = PRECOND[precond]
    p.n_overlap = n_overlap
    p.weighting_type = WEIGHTING[weighting]
    p.element_centric = 1 if element_centric else 0
    p.cheb_degree = cheb_degree
    p.smoothing_range = smoothing_range
    p.eig_n_iterations = eig_n_iterations
    p.ev_algorithm = EV[ev_algorithm]
    p.kernel_variant = KERNEL[kernel]
    p.device = device
    p.stream = stream
    p.rank = rank
    p.world_size = world_size
    p.polynomial_type = POLYNOMIAL[polynomial_type]
    # options: {reference JSON key: string value}, applied last through asm_params_set
    for key, value in (options or {}).items():
        if lib.asm_params_set(C.byref(p), str(key).encode(), str(value).encode()):
            raise ValueError(lib.asm_last_error(None).decode())
    p._keepalive = keep
    return p


def describe_layout(params: AsmParams) -> AsmLayout:
    lib = load_library()
    out = AsmLayout()
    err = lib.asm_describe_layout(C.byref(params), C.byref(out))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    return out


def exchange_plan_direct(params: AsmParams):
    """list of asm_direct_message: the single-round form of update_ghost_values"""
    lib = load_library()
    msgs = (AsmDirectMessage * 26)()
    n = C.c_int32()
    err = lib.asm_exchange_plan_direct(C.byref(params), msgs, C.byref(n))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    return [msgs[i] for i in range(n.value)]


def exchange_plan(params: AsmParams, add: bool):
    """list of asm_exchange_step of update_ghost_values (add=False) / compress(add)"""
    lib = load_library()
    steps = (AsmExchangeStep * 3)()
    n = C.c_int32()
    err = lib.asm_exchange_plan(C.byref(params), 1 if add else 0, steps, C.byref(n))
    if err:
        raise ValueError(lib.asm_last_error(None).decode())
    return [steps[i] for i in range(n.value)]


def loopback_init(contexts):
    """Testing aid (asm_debug_loopback_init): contexts[r] = rank r; drive every rank from its own thread."""
    lib = load_library()
    n = len(contexts)
    cs = (C.c_void_p * n)(*[c.handle for c in contexts])
    err = lib.asm_debug_loopback_init(cs, n)
    if err:
        raise RuntimeError(f"asm_debug_loopback_init failed ({err}): {lib.asm_last_error(contexts[0].handle).decode()}")


def exchange_all(contexts, vectors, add=False):
    """Testing aid (asm_debug_exchange_all): ghost exchange of vectors[r] of contexts[r], r = rank,
    for all ranks of a partition living in this process."""
    lib = load_library()
    n = len(contexts)
    cs = (C.c_void_p * n)(*[c.handle for c in contexts])
    vs = (C.c_void_p * n)(*[v.ptr for v in vectors])
    err = lib.asm_debug_exchange_all(cs, vs, n, 1 if add else 0)
    if err:
        raise RuntimeError(f"asm_debug_exchange_all failed ({err}): {lib.asm_last_error(contexts[0].handle).decode()}")


class DeviceVector:
    """A device buffer in the library's box layout (initialize_dof_vector)."""

    def __init__(self, ctx: "Context"):
        self.ctx = ctx
        ptr = C.c_void_p()
        ctx._check(ctx.lib.asm_vector_alloc(ctx.handle, C.byref(ptr)))
        self.ptr = ptr

    def set(self, host_owned):
        a = np.ascontiguousarray(host_owned, dtype=np.float64).reshape(-1)
        assert a.size == self.ctx.layout.n_owned, (a.size, self.ctx.layout.n_owned)
        self.ctx._check(self.ctx.lib.asm_vector_import(self.ctx.handle, self.ptr, a.ctypes.data_as(C.c_void_p)))
        return self

    def get(self):
        out = np.empty(self.ctx.layout.n_owned, dtype=np.float64)
        self.ctx._check(self.ctx.lib.asm_vector_export(self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.ptr))
        return out

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            self.ctx.lib.asm_vector_free(self.ctx.handle, self.ptr)
        self.ptr = None


class Context:
    """Operator + preconditioner + Chebyshev smoother on one GPU."""

    def __init__(self, params: AsmParams | None = None, _borrowed_handle=None, **kw):
        self.lib = load_library()
        self._owned = _borrowed_handle is None
        if _borrowed_handle is not None:  # a level of a Multigrid object: owned by it
            self.params = None
            self.handle = _borrowed_handle
        else:
            self.params = params if params is not None else make_params(**kw)
            h = C.c_void_p()
            err = self.lib.asm_create(C.byref(self.params), C.byref(h))
            if err:
                raise RuntimeError(f"asm_create failed ({err}): {self.lib.asm_last_error(None).decode()}")
            self.handle = h
        self.layout = AsmLayout()
        self._check(self.lib.asm_vector_layout(self.handle, C.byref(self.layout)))

    def _check(self, err):
        if err:
            raise RuntimeError(f"dealii_asm_amd error {err}: {self.lib.asm_last_error(self.handle).decode()}")

    def close(self):
        if self.handle is not None and self._owned:
            self.lib.asm_destroy(self.handle)
        self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- vector algebra (LinearAlgebra::distributed::Vector operations of the solvers) -----------
    def vec_dot(self, x, y):
        out = C.c_double()
        self._check(self.lib.asm_vec_dot(self.handle, x.ptr, y.ptr, C.byref(out)))
        return out.value

    def vec_multi_dot(self, w, vs):
        """[w * v for v in vs] in one sweep over the vectors"""
        n = len(vs)
        ptrs = (C.c_void_p * max(n, 1))(*[v.ptr for v in vs])
        out = (C.c_double * max(n, 1))()
        self._check(self.lib.asm_vec_multi_dot(self.handle, w.ptr, ptrs, n, out))
        return [out[i] for i in range(n)]

    def vec_multi_axpy_norm(self, w, vs, a):
        """w += sum_i a[i] vs[i]; returns w * w afterwards"""
        n = len(vs)
        ptrs = (C.c_void_p * max(n, 1))(*[v.ptr for v in vs])
        coef = (C.c_double * max(n, 1))(*[float(x) for x in a])
        out = C.c_double()
        self._check(self.lib.asm_vec_multi_axpy_norm(self.handle, w.ptr, ptrs, coef, n, C.byref(out)))
        return out.value

    def vec_axpby(self, y, a, x, b):
        self._check(self.lib.asm_vec_axpby(self.handle, y.ptr, float(a), x.ptr, float(b)))

    def vec_scale(self, y, a, x):
        self._check(self.lib.asm_vec_scale(self.handle, y.ptr, float(a), x.ptr))

    def vec_copy(self, y, x):
        self._check(self.lib.asm_vec_copy(self.handle, y.ptr, x.ptr))

    def vec_zero(self, y):
        self._check(self.lib.asm_vec_zero(self.handle, y.ptr))

    def constraints_set_zero(self, y):
        self._check(self.lib.asm_vec_constraints_set_zero(self.handle, y.ptr))

    def solve(self, x, b, solver="GMRES", preconditioner="Chebyshev", mg=None, rel_tol=1e-2, abs_tol=1e-10, max_iter=1000,
              right_preconditioning=True, max_n_tmp_vectors=30, keep_preconditioned_basis=False):
        """solve() of element_centered_preconditioners_01.cc:108-263 (x = 0 initially); returns AsmSolverResult"""
        sp = AsmSolverParams()
        self.lib.asm_solver_params_default(C.byref(sp))
        sp.type = SOLVER[solver]
        sp.rel_tolerance, sp.abs_tolerance, sp.max_iterations = rel_tol, abs_tol, max_iter
        sp.right_preconditioning = 1 if right_preconditioning else 0
        sp.max_n_tmp_vectors = max_n_tmp_vectors
        sp.keep_preconditioned_basis = 1 if keep_preconditioned_basis else 0
        res = AsmSolverResult()
        err = self.lib.asm_solve(self.handle, mg.handle if mg is not None else None, SYSTEM_PRECONDITIONER[preconditioner],
                                 C.byref(sp), x.ptr, b.ptr, C.byref(res))
        if err:
            msg = self.lib.asm_mg_last_error(mg.handle).decode() if mg is not None else self.lib.asm_last_error(self.handle).decode()
            raise RuntimeError(f"asm_solve failed ({err}): {msg}")
        return res

    # -- vectors ------------------------------------------------------------
    def vector(self, host_owned=None):
        v = DeviceVector(self)
        if host_owned is not None:
            v.set(host_owned)
        return v

    @property
    def n_owned(self):
        return int(self.layout.n_owned)

    @property
    def n_dofs(self):
        return int(self.layout.n_global_dofs)

    def set_keep_ghosts(self, on=True):
        """Promise: vectors are only modified through the library between calls (asm_set_keep_ghosts)."""
        self._check(self.lib.asm_set_keep_ghosts(self.handle, 1 if on else 0))

    def two_phase(self, on=True):
        """Testing aid: tiled sweeps as boundary + interior launches (asm_debug_two_phase)."""
        self._check(self.lib.asm_debug_two_phase(self.handle, 1 if on else 0))

    def external_exchange(self, on=True):
        """Testing aid: vmult/step leave the ghost exchange to exchange_all()."""
        self._check(self.lib.asm_debug_external_exchange(self.handle, 1 if on else 0))

    # -- operators (names follow the reference) -------------------------------
    def op_vmult(self, dst, src):
        self._check(self.lib.asm_op_vmult(self.handle, dst.ptr, src.ptr))

    def op_inverse_diagonal(self, out):
        self._check(self.lib.asm_op_inverse_diagonal(self.handle, out.ptr))

    def precond_vmult(self, dst, src):
        self._check(self.lib.asm_precond_vmult(self.handle, dst.ptr, src.ptr))

    def op_vmult_fused(self, mode, dst, src, b=None):
        """LaplaceOperatorMatrixFree::vmult(dst, src, pre, post) for the updates of asm_fused_update"""
        self._check(self.lib.asm_op_vmult_fused(self.handle, FUSED[mode], dst.ptr, src.ptr, b.ptr if b is not None else None))

    def precond_vmult_fused(self, mode, dst, src, x=None, x_old=None, f1=0.0, f2=1.0):
        """ASPoissonPreconditioner::vmult(dst, src, pre, post) for the updates of asm_fused_update"""
        self._check(
            self.lib.asm_precond_vmult_fused(
                self.handle, FUSED[mode], dst.ptr, src.ptr, x.ptr if x is not None else None,
                x_old.ptr if x_old is not None else None, float(f1), float(f2)
            )
        )

    def is_symmetric(self):
        v = C.c_int32()
        self._check(self.lib.asm_is_symmetric(self.handle, C.byref(v)))
        return bool(v.value)

    def estimate_eigenvalues(self):
        lam = C.c_double()
        self._check(self.lib.asm_cheb_estimate_eigenvalues(self.handle, C.byref(lam)))
        return lam.value

    def set_eigenvalue(self, lam):
        self._check(self.lib.asm_cheb_set_eigenvalue(self.handle, float(lam)))

    def cheb_vmult(self, dst, src):
        self._check(self.lib.asm_cheb_vmult(self.handle, dst.ptr, src.ptr))

    def cheb_step(self, dst, src):
        self._check(self.lib.asm_cheb_step(self.handle, dst.ptr, src.ptr))

    def update_ghost_values(self, v):
        self._check(self.lib.asm_update_ghost_values(self.handle, v.ptr))

    def compress_add(self, v):
        self._check(self.lib.asm_compress_add(self.handle, v.ptr))

    def synchronize(self):
        self._check(self.lib.asm_synchronize(self.handle))

    def n_fdm_instances(self):
        n = C.c_int64()
        self._check(self.lib.asm_n_fdm_instances(self.handle, C.byref(n)))
        return n.value

    def memory_consumption(self):
        n = C.c_int64()
        self._check(self.lib.asm_memory_consumption(self.handle, C.byref(n)))
        return n.value

    def cheb_parameters(self):
        out = (C.c_double * 3)()
        self._check(self.lib.asm_cheb_parameters(self.handle, out))
        return {"theta": out[0], "delta": out[1], "lambda_max": out[2]}

    def kernel_variant(self):
        v = C.c_int32()
        self._check(self.lib.asm_kernel_variant_in_use(self.handle, C.byref(v)))
        return {1: "generic", 2: "tiled"}.get(v.value, str(v.value))

    def time_calls(self, what, reps, dst, src):
        """HIP-event time (ms) of `reps` back-to-back calls; what in
        op_vmult|precond_vmult|cheb_vmult|cheb_step."""
        idx = {"op_vmult": 0, "precond_vmult": 1, "cheb_vmult": 2, "cheb_step": 3}[what]
        ms = C.c_float()
        self._check(self.lib.asm_time_calls(self.handle, idx, reps, dst.ptr, src.ptr, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._check(self.lib.asm_profile_enable(self.handle, 1 if on else 0))

    def profile_get(self, kernel):
        """(launches, total ms) of kernel 'laplace' or 'fdm' since profile_enable"""
        n, ms = C.c_int64(), C.c_double()
        self._check(self.lib.asm_profile_get(self.handle, {"laplace": 0, "fdm": 1}[kernel], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def measure_copy_bandwidth(self, bytes_per_array=1 << 30, reps=5):
        """GB/s of a plain device-to-device copy (read + written bytes), asm_measure_copy_bandwidth"""
        g = C.c_double()
        self._check(self.lib.asm_measure_copy_bandwidth(self.handle, bytes_per_array, reps, C.byref(g)))
        return g.value

    def pair_sweeps(self):
        """(laplace, fdm): which sweeps run on the wave-pair kernels"""
        a, f = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.asm_pair_sweeps_in_use(self.handle, C.byref(a), C.byref(f)))
        return bool(a.value), bool(f.value)

    def pair_zchunks(self):
        """z chunks of the wave-pair sweeps (0: the schedule model's first choice)"""
        z = C.c_int32(0)
        self._check(self.lib.asm_pair_zchunks(self.handle, C.byref(z)))
        return z.value

    def pair_waves(self):
        """waves per tile of the wave-pair sweeps (4 or 2)"""
        w = C.c_int32(0)
        self._check(self.lib.asm_pair_waves(self.handle, C.byref(w)))
        return w.value

    def tiled_laplace_rows(self):
        r = C.c_int32()
        self._check(self.lib.asm_tiled_laplace_rows(self.handle, C.byref(r)))
        return r.value

    def comm_init(self, unique_id: bytes):
        self._check(self.lib.asm_comm_init(self.handle, unique_id))


class Multigrid:
    """Level hierarchy + V-cycle (PreconditionerGMG, include/multigrid.h:40-520) over contexts of this library."""

    def __init__(self, fine_params: AsmParams, mg_type="h", p_sequence="bisect", one_sided=False, coarse="auto",
                 coarse_rel_tol=1e-3, coarse_max_iter=200, estimate_eigenvalues=True, n_coarse_cycles=1,
                 intermediate_smoother=None, keep_duplicate_level=True):
        """intermediate_smoother: None ("mg intermediate smoother" without a type = the mg smoother) or
        (inner preconditioner "FDM" | "Diagonal", Chebyshev degree)"""
        self.lib = load_library()
        mp = AsmMgParams()
        self.lib.asm_mg_params_default(C.byref(mp))
        mp.n_coarse_cycles = n_coarse_cycles
        mp.keep_duplicate_level = 1 if keep_duplicate_level else 0
        if intermediate_smoother is not None:
            mp.intermediate_smoother = 1
            mp.intermediate_precond_type = PRECOND[intermediate_smoother[0]]
            mp.intermediate_cheb_degree = int(intermediate_smoother[1])
        mp.mg_type, mp.p_sequence = MG_TYPE[mg_type], MG_P_SEQUENCE[p_sequence]
        mp.one_sided_v_cycle = 1 if one_sided else 0
        mp.coarse_solver = MG_COARSE[coarse]
        mp.coarse_rel_tolerance, mp.coarse_max_iterations = coarse_rel_tol, coarse_max_iter
        mp.estimate_eigenvalues = 1 if estimate_eigenvalues else 0
        h = C.c_void_p()
        err = self.lib.asm_mg_create(C.byref(fine_params), C.byref(mp), C.byref(h))
        if err:
            raise RuntimeError(f"asm_mg_create failed ({err}): {self.lib.asm_mg_last_error(None).decode()}")
        self.handle = h
        n = C.c_int32()
        self.lib.asm_mg_n_levels(h, C.byref(n))
        self.n_levels = n.value
        self.levels = []
        for l in range(self.n_levels):
            ch = C.c_void_p()
            self._check(self.lib.asm_mg_level_context(h, l, C.byref(ch)))
            self.levels.append(Context(_borrowed_handle=ch))

    def level_distributed(self, level):
        d = C.c_int32()
        self._check(self.lib.asm_mg_level_distributed(self.handle, level, C.byref(d)))
        return bool(d.value)

    def graph_replays(self):
        """(level, replays) of the hipGraph of the small levels' legs; level -1: no graph"""
        level, n = C.c_int32(), C.c_int64()
        self._check(self.lib.asm_mg_graph_replays(self.handle, C.byref(level), C.byref(n)))
        return level.value, n.value

    def comm_init(self, unique_id: bytes):
        """asm_mg_comm_init: one RCCL communicator for all block-decomposed levels (collective)"""
        self._check(self.lib.asm_mg_comm_init(self.handle, unique_id))

    def after_loopback(self):
        """testing aid (asm_mg_debug_after_loopback), from this rank's own thread"""
        self._check(self.lib.asm_mg_debug_after_loopback(self.handle))

    def _check(self, err):
        if err:
            raise RuntimeError(f"dealii_asm_amd multigrid error {err}: {self.lib.asm_mg_last_error(self.handle).decode()}")

    @property
    def fine(self):
        return self.levels[-1]

    def prolongate_add(self, level, dst_fine, src_coarse):
        self._check(self.lib.asm_mg_prolongate_add(self.handle, level, dst_fine.ptr, src_coarse.ptr))

    def restrict(self, level, dst_coarse, src_fine):
        self._check(self.lib.asm_mg_restrict(self.handle, level, dst_coarse.ptr, src_fine.ptr))

    def vmult(self, dst, src):
        self._check(self.lib.asm_mg_vmult(self.handle, dst.ptr, src.ptr))

    def close(self):
        if self.handle is not None:
            for c in self.levels:
                c.handle = None
            self.lib.asm_mg_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    if lib.asm_comm_unique_id(buf):
        raise RuntimeError("ncclGetUniqueId failed")
    return buf.raw
