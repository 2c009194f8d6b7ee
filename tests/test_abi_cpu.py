"""CPU-side checks of the C ABI: the in-tree library loads, exports every symbol the
header declares, describes layouts / exchange plans without a GPU, and refuses to
run (instead of falling back to the CPU) when no HIP device is present."""
import itertools
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(pkg):
    header = open(os.path.join(ROOT, "include", "dealii_asm_amd.h")).read()
    declared = set(re.findall(r"\b(asm_[a-z_0-9]+)\s*\(", header))
    lib = pkg.load_library()
    assert declared == set(pkg.EXPORTED_SYMBOLS), declared ^ set(pkg.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_header_cites_reference_interfaces():
    header = open(os.path.join(ROOT, "include", "dealii_asm_amd.h")).read()
    for cite in ("include/operator.h:1353-1365", "include/matrix_free.h:925-954", "include/matrix_free_internal.h"):
        assert cite in header


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("boundary", ["dirichlet", "periodic"])
@pytest.mark.parametrize("kernel", ["generic", "tiled"])
def test_layout_partitions_all_dofs(pkg, world, boundary, kernel):
    nc, p = (8, 12, 16), 4
    n_glob = 1
    for d in range(3):
        n_glob *= nc[d] * p + (0 if boundary == "periodic" else 1)
    owned = np.zeros([nc[d] * p + 1 for d in range(3)], dtype=int)
    total = 0
    for rank in range(world):
        prm = pkg.make_params(degree=p, n_cells=nc, boundary=boundary, world_size=world, rank=rank, kernel=kernel)
        L = pkg.describe_layout(prm)
        assert L.n_global_dofs == n_glob
        sl = tuple(slice(L.owned_begin[d], L.owned_begin[d] + L.owned_size[d]) for d in range(3))
        owned[sl] += 1
        total += L.n_owned
        assert L.n_alloc >= L.n_owned + L.n_ghost
        assert L.stride[1] % 8 == 0  # padded rows: 64-byte aligned row starts
        assert all(L.store_begin[d] <= 0 for d in range(3))
        assert np.prod([L.proc_grid[d] for d in range(3)]) == world
    assert total == n_glob
    glob = tuple(slice(0, nc[d] * p + (0 if boundary == "periodic" else 1)) for d in range(3))
    assert np.all(owned[glob] == 1)


def test_exchange_plans_are_consistent(pkg):
    """what one rank sends is what its peer expects to receive, for every patch type"""
    cases = [dict(n_overlap=1, kernel="generic"), dict(n_overlap=2, kernel="generic"), dict(element_centric=False, kernel="generic"),
             dict(n_overlap=1, kernel="tiled")]
    for kw, world, boundary in itertools.product(cases, [2, 8], ["dirichlet", "periodic"]):
        plans = {}
        for add in (False, True):
            for rank in range(world):
                prm = pkg.make_params(degree=4, n_cells=(8, 8, 8), boundary=boundary, world_size=world, rank=rank, **kw)
                plans[(add, rank)] = pkg.exchange_plan(prm, add)
        for (add, rank), steps in plans.items():
            for st in steps:
                for side in (0, 1):
                    peer = st.peer[side]
                    n_send = int(np.prod(list(st.send_size[side]))) if peer >= 0 else 0
                    if n_send == 0:
                        continue
                    other = [s for s in plans[(add, peer)] if s.direction == st.direction][0]
                    # my low neighbour receives it from its high side and vice versa
                    assert other.peer[1 - side] == rank
                    assert list(other.recv_size[1 - side]) == list(st.send_size[side])


def test_invalid_parameters_are_reported(pkg):
    bad = pkg.make_params(degree=9)
    with pytest.raises(ValueError, match="degree"):
        pkg.describe_layout(bad)
    bad = pkg.make_params(boundary=[("periodic", "dirichlet")] * 3)
    with pytest.raises(ValueError, match="periodic"):
        pkg.describe_layout(bad)
    bad = pkg.make_params(degree=3, kernel="tiled")
    with pytest.raises(ValueError, match="tiled"):
        pkg.describe_layout(bad)


def test_no_cpu_fallback(pkg):
    import ctypes

    lib = pkg.load_library()
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        pkg.Context(n_cells=(2, 2, 2))


def test_params_set_accepts_reference_keys_and_rejects_unknown(pkg):
    """asm_params_set: the reference's JSON keys (precondition.templates.h:108-152, 195-211); unknown keys and
    values are errors, as the reference throws for them (:26, 130-133, 148-152)"""
    import ctypes as C

    lib = pkg.load_library()
    p = pkg.AsmParams()
    lib.asm_params_default(C.byref(p))
    for key, value in [("n overlap", "2"), ("weighting type", "ras"), ("element centric", "false"), ("type", "Diagonal"),
                       ("degree", "5"), ("smoothing range", "15.5"), ("ev algorithm", "power iteration"),
                       ("polynomial type", "4th kind"), ("fe degree", "7")]:
        assert lib.asm_params_set(C.byref(p), key.encode(), value.encode()) == 0, key
    assert (p.n_overlap, p.weighting_type, p.element_centric, p.precond_type) == (2, pkg.WEIGHTING["ras"], 0, pkg.PRECOND["Diagonal"])
    assert (p.cheb_degree, p.smoothing_range, p.ev_algorithm, p.polynomial_type, p.degree) == (5, 15.5, 2, 1, 7)
    for key, value in [("weighting type", "both"), ("polynomial type", "2nd kind"), ("ev algorithm", "qr"), ("no such key", "1"),
                       ("degree", "three")]:
        assert lib.asm_params_set(C.byref(p), key.encode(), value.encode()) != 0, (key, value)
        assert "not known" in lib.asm_last_error(None).decode() or "unknown option key" in lib.asm_last_error(None).decode()
    # the reference's implementation-choice keys (precondition.templates.h:195-211, 341, 388, 534): every value the
    # reference takes is accepted and leaves the parameters unchanged, so its JSON block can be forwarded verbatim
    before = bytes(p)
    for key, value in [("weight sequence", "global"), ("weight sequence", "compressed"), ("weight sequence", "dg"),
                       ("reuse partitioner", "true"), ("reuse partitioner", "false"), ("overlap pre post", "false"),
                       ("optimize", "0"), ("optimize", "3"), ("sub mesh approximation", "3")]:
        assert lib.asm_params_set(C.byref(p), key.encode(), value.encode()) == 0, (key, value)
    assert bytes(p) == before
    for key, value in [("weight sequence", "local"), ("reuse partitioner", "maybe"), ("optimize", "4"), ("sub mesh approximation", "0")]:
        assert lib.asm_params_set(C.byref(p), key.encode(), value.encode()) == 1, (key, value)
    assert lib.asm_params_set(C.byref(p), b"sub mesh approximation", b"1") == 2
    assert "not implemented" in lib.asm_last_error(None).decode()
    # the Chebyshev parameter checks of the layout setup
    q = pkg.make_params(cheb_degree=0)
    with pytest.raises(ValueError):
        pkg.describe_layout(q)
    q = pkg.make_params(smoothing_range=1.0)
    with pytest.raises(ValueError):
        pkg.describe_layout(q)


def test_comm_init_error_path(pkg):
    """asm_comm_init without a context / id is an error code, not a crash (the only part of the RCCL bootstrap that
    can be walked without a GPU)"""
    import ctypes as C

    lib = pkg.load_library()
    assert lib.asm_comm_init(None, b"\0" * 128) == 1
    assert "context and unique id" in lib.asm_last_error(None).decode()


def test_params_of_another_abi_revision_are_refused(pkg):
    """asm_params carries its size (set by asm_params_default): a structure that was not filled by it - a caller built
    against another revision of the header - is refused by every entry point that takes one (no GPU needed for this one)"""
    import ctypes as C

    lib = pkg.load_library()
    p = pkg.AsmParams()
    lib.asm_params_default(C.byref(p))
    assert p.struct_size == C.sizeof(pkg.AsmParams)
    layout = pkg.AsmLayout()
    assert lib.asm_describe_layout(C.byref(p), C.byref(layout)) == 0
    p.struct_size = C.sizeof(pkg.AsmParams) - 4  # revision 2 ended before this field
    assert lib.asm_describe_layout(C.byref(p), C.byref(layout)) == 1
    assert "ABI revision" in lib.asm_last_error(None).decode()
    sp = pkg.AsmSolverParams()
    lib.asm_solver_params_default(C.byref(sp))
    assert sp.struct_size == C.sizeof(pkg.AsmSolverParams)


@pytest.mark.parametrize("rows", [0, 9, 10, 11])
@pytest.mark.parametrize("nc,bc", [((80, 80, 80), "dirichlet"), ((78, 78, 78), "dirichlet"), ((128, 128, 96), "dirichlet"), ((128, 64, 64), "periodic"),
                                   ((12, 8, 2), "dirichlet"), ((13, 9, 3), "dirichlet"), ((23, 15, 40), ["dirichlet", "periodic", "neumann"]),
                                   ((1, 1, 1), "dirichlet"), ((100, 7, 33), ["periodic", "neumann", "dirichlet"])])
def test_tiling_of_the_tiled_sweeps(pkg, nc, bc, rows):
    """asm_tiled_plan (host only): the tiles of the degree-4 owner-computes sweeps own every cell column / row of the block
    exactly once - tile t covers cells t * owned + first ... + owned and owns all but its first, which is its halo unless
    nothing lies below it (tile 0 at a physical boundary: first = 0) -, the z chunks cover the layers once, and the work
    order is a permutation that keeps every XCD's contiguous range of the work (kernels_tiled.hip: make_tiling, tiled_order)"""
    prm = pkg.make_params(degree=4, n_cells=nc, boundary=bc)
    plan, order = pkg.tiled_plan(prm, n_cu=256, rows=rows)
    bcs = [bc] * 3 if isinstance(bc, str) else bc
    assert plan["owned_y"] == (7 if rows == 0 else rows - 1) and plan["owned_x"] == 11
    for d, (n_tiles, owned, first) in enumerate(((plan["ntx"], plan["owned_x"], plan["x_first"]), (plan["nty"], plan["owned_y"], plan["y_first"]))):
        assert first == (-1 if bcs[d] == "periodic" else 0)
        count = [0] * nc[d]
        for t in range(n_tiles):
            lo = t * owned + first
            for c in range(lo, lo + owned + 1):
                halo = c == lo and not (t == 0 and first == 0)
                if 0 <= c < nc[d] and not halo:
                    count[c] += 1
            assert lo <= nc[d] - 1  # no tile beyond the block
        assert count == [1] * nc[d], (d, count)
    assert plan["nzc"] * plan["zlen"] >= nc[2] > (plan["nzc"] - 1) * plan["zlen"]
    n = plan["ntx"] * plan["nty"] * plan["nzc"]
    if order:
        assert len(order) == n and sorted(order) == list(range(n))
        for x in range(8):  # (xcd_remap: XCD x takes work indices [begin, begin + size))
            begin, size = x * (n >> 3) + min(x, n & 7), (n >> 3) + (1 if x < (n & 7) else 0)
            assert sorted(order[begin:begin + size]) == list(range(begin, begin + size))
    else:
        assert all(b != "dirichlet" for b in bcs) or n < 16


def test_chunk_count_model_of_the_tiled_sweeps(pkg):
    """the chunk count of a block with Dirichlet faces is never ONE chunk of the whole height when that leaves the launch
    without a second workgroup per compute unit to balance with (round 2's model took 1 chunk of 96 layers for
    128 x 128 x 96 cells); a periodic block keeps the rounds-only model (two chunks, 240 workgroups in one round)"""
    plan, _ = pkg.tiled_plan(pkg.make_params(degree=4, n_cells=(128, 128, 96), boundary="dirichlet"))
    assert 4 <= plan["nzc"] <= 16
    plan, _ = pkg.tiled_plan(pkg.make_params(degree=4, n_cells=(80, 80, 80), boundary="dirichlet"))
    assert plan["nzc"] == 8 and plan["zlen"] == 10
    plan, _ = pkg.tiled_plan(pkg.make_params(degree=4, n_cells=(128, 64, 64), boundary="periodic"))
    assert plan["nzc"] == 2


@pytest.mark.parametrize("nc,boundary", [((80, 80, 80), "dirichlet"), ((128, 128, 96), "dirichlet"), ((7, 8, 3), "dirichlet"), ((15, 9, 4), "neumann"),
                                         ((16, 16, 9), "periodic"), ((1, 1, 1), "dirichlet"), ((23, 57, 5), "dirichlet")])
def test_pair_plan_covers_the_block(pkg, nc, boundary):
    """asm_pair_plan / asm_pair_tiles (host only): the tiles of the wave-pair sweeps - 8 x 8 cells of which 7 x 7 are owned (all 8
    where a tile starts at cell 0 of a physical boundary), strip tiles of 4 x 16 / 16 x 4 cells where the last tile column / row
    would own three cells or less - own every cell of the block exactly once, plus the cell behind the last one where the
    block owns its last DoF plane; the z chunks cover the layers; the work order is a permutation that keeps every XCD's
    contiguous range"""
    prm = pkg.make_params(degree=4, n_cells=nc, boundary=boundary)
    plan, order = pkg.pair_plan(prm, n_cu=256)
    tiles = pkg.pair_tiles(prm, n_cu=256)
    periodic = boundary == "periodic"
    assert plan["x_first"] == plan["y_first"] == (-1 if periodic else 0)
    last = [nc[d] - 1 if periodic else nc[d] for d in range(2)]  # non-periodic: the cell nc owns the plane 4 nc
    owned = np.zeros((last[1] + 1, last[0] + 1), dtype=int)
    for x0, y0, w, h, lo, x_max in tiles:
        assert (w, h) in ((8, 8), (4, 16), (16, 4))
        xs = [c for c in range(x0, x0 + w) if 0 <= c <= min(last[0], x_max) and (c > x0 or lo & 1)]
        ys = [c for c in range(y0, y0 + h) if 0 <= c <= last[1] and (c > y0 or lo & 2)]
        assert xs and ys, "a tile that owns nothing"
        owned[np.ix_(ys, xs)] += 1
    assert (owned == 1).all(), owned
    assert len(tiles) <= plan["ntx"] * plan["nty"] + ((last[1] + 15) // 15 + 1) + ((last[0] + 15) // 15 + 1)
    full = [-(-(last[d] + (1 if plan["x_first"] else 0) - (0 if periodic else 1)) // 7) for d in range(2)]
    assert len(tiles) <= (full[0] + 1) * (full[1] + 1)
    assert plan["nzc"] * plan["zlen"] >= nc[2] and (plan["nzc"] - 1) * plan["zlen"] < nc[2]
    n = len(tiles) * plan["nzc"]
    assert len(order) == n and sorted(order) == list(range(n))
    for xcd in range(8):
        begin, count = xcd * (n // 8) + min(xcd, n % 8), n // 8 + (1 if xcd < n % 8 else 0)
        assert sorted(order[begin:begin + count]) == list(range(begin, begin + count))


def test_pair_strip_tiles_at_the_bench_sizes(pkg):
    """80^3 and 128 x 128 x 96 cells: 81 = 8 + 10 x 7 + 3 and 129 = 8 + 17 x 7 + 2 cell columns - strips replace the last tile
    column and row (133 instead of 144 tiles, 342 instead of 361)"""
    for nc, n_plain, n_strips in (((80, 80, 80), 144, 11 * 11 + 6 + 6), ((128, 128, 96), 361, 18 * 18 + 9 + 9)):
        tiles = pkg.pair_tiles(pkg.make_params(degree=4, n_cells=nc, boundary="dirichlet"))
        assert len(tiles) == n_strips < n_plain, len(tiles)
