#!/usr/bin/env python3
"""Benchmark of the hot path: one Chebyshev-wrapped FDM additive-Schwarz smoother
application (PreconditionChebyshev::step, as timed by the reference harness
matrix_free_loop_08.likwid.cc:345-395), 3-D, p=4, on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU;
the mesh is block-decomposed (weak scaling: every rank keeps `--cells` cells) and
ghost planes travel over RCCL inside the library.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  cart_dirichlet  Cartesian box, all faces Dirichlet (experiments/amd-opt_fdm1_cart.json:
                  Kershaw generator with eps=1), default 80^3 cells/GPU = 33 076 161 DoFs
  cart_periodic   periodic box of the reference harness (matrix_free_loop_08.likwid.cc:160-174)
  kershaw         Kershaw eps=0.3 (experiments/large-scaling-opt_fdm1.json), Q2 geometry
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6  # MI355X vector fp64 peak (256 CUs x 4 SIMDs x 16 FMA lanes x 2.4 GHz x 2)
# fp64 operations per DoF of the deformed-cell (tri-quadratic) Laplace sweep, degree 4 (sum-factorised
# evaluate + integrate 2 x 6 x 2 n1^4 per cell, Jacobian from 27 coefficient vectors, 3 x 3 inverse and two 3 x 3 products
# per quadrature point; an estimate of the arithmetic the algorithm needs, not a count of issued instructions)
KERSHAW_LAPLACE_FLOP_PER_DOF = 450.0
NORTH_STAR_CELLS = (128, 128, 96)  # 100 860 545 DoFs at p = 4: the ~1e8-DoF point the north star is quoted on


def algorithmic_bytes_per_dof(degree, what):
    """fp64 vector bytes per DoF that one smoother application must move (SURVEY.md 8d,
    metadata excluded: this implementation stores no index lists).
      A residual sweep: read x, read b, write r           = 24
      P sweep, first (x0 = 0): read b, write x            = 16
      P sweep after a residual: read r, x(, x_old), write = 24 / 32
    """
    a_sweep = 24.0
    if what == "cheb_vmult":
        p_sweeps = [16.0] + [24.0 if j == 1 else 32.0 for j in range(1, degree)]
        a_sweeps = [a_sweep] * (degree - 1)
    else:
        p_sweeps = [24.0] + [32.0] * (degree - 1)
        a_sweeps = [a_sweep] * degree
    return a_sweeps, p_sweeps


def cpu_baseline(args, pkg, cells):
    """The same smoother application on the host cores, on a bounded sample of the workload.
    The threaded C++ restatement oracle/asm_oracle_cpu.cc (kind "port", checked against the numpy oracle in
    tests/test_oracle_cpu_port.py) on every host core this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asm_oracle as orc

    # host cores this process may use (the GPU box gives one GPU's CPU share) vs what the node has
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # ... and the CPU time the control group grants (cgroup v2 cpu.max "quota period"): a GPU box shows all 256 hardware
    # threads of its host but is throttled to 16 CPUs' worth of time, so that more threads than that run SLOWER
    # (profiles/r03_cpu_baseline_threads.txt: 3.9e8 DoF/s at 16 threads, 1.6e8 at 256)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(q) // int(period))
    except (OSError, ValueError):
        pass
    granted = min(usable, quota) if quota else usable
    if args.workload != "kershaw":
        import asm_oracle_cpu

        threads = max(1, granted if args.cpu_threads <= 0 else min(usable, args.cpu_threads))
        # fewer threads than hardware threads: one per physical core, spread over the host's core complexes
        cpus = asm_oracle_cpu.spread_cpus(threads) if (threads < usable and os.environ.get("ASM_BENCH_NO_PIN") is None) else None
        if cpus:
            os.environ["ASM_ORACLE_CPUS"] = ",".join(str(c) for c in cpus)
        nc = tuple(cells)  # the mesh of the GPU measurement itself
        lib = asm_oracle_cpu.native_library()  # built here, on the host it runs on, with -march=native
        native = "native" in os.path.basename(lib)
        sm = asm_oracle_cpu.CpuSmoother(nc, args.workload == "cart_periodic", degree=args.degree, lambda_max=1.7, n_threads=threads, lib=lib)
        # each call sets up its thread pool and first-touches its work vectors: several steps per call
        b = np.ones(sm.n_dofs)
        x = sm.step(np.zeros(sm.n_dofs), b, reps=2)  # warm-up
        per_call = 8
        reps, t0 = 0, time.perf_counter()
        while True:
            x = sm.step(x, b, reps=per_call)
            reps += per_call
            el = time.perf_counter() - t0
            if el > 12.0 or reps >= 400:
                break
        return {
            "value": sm.n_dofs * reps / el,
            "unit": "DoF/s",
            "cores": threads,
            "kind": "port",
            "host_cpu_count": os.cpu_count(),
            "host_cpus_usable": usable,
            "host_cpu_quota": quota,
            "pinned_cpus": cpus,
            "sample": f"{reps} Chebyshev(k={args.degree})+FDM step() on {nc[0]}x{nc[1]}x{nc[2]} cells p=4 "
            f"({sm.n_dofs} DoFs, the bench mesh), C++ restatement oracle/asm_oracle_cpu.cc "
            f"({'-march=native' if native else '-march=x86-64-v3'}; blocks of 8x4x4 cells per thread and colour, SIMD batches of 8 cells), "
            f"{threads} threads (os.cpu_count()={os.cpu_count()}, usable={usable}, cgroup cpu quota={quota})",
            # the reference's own published solve (data/data.pdf Table 8: Kershaw eps=1 p=4, 7.19e6 DoFs, 4 GMRES iterations,
            # 9.52e-2 s on 40 MPI ranks, Chebyshev k=1) implies, with ~2 fine-level smoother steps per V-cycle plus the
            # coarser levels, roughly 1e9 DoF/s per fine-level smoother step: a DERIVED figure for other hardware, quoted
            # for scale only
            "reference_derived": {"value": 1.0e9, "unit": "DoF/s", "basis": "data/data.pdf Table 8, 40 MPI ranks ('gold'), derived"},
        }
    # Kershaw: the threaded C++ restatement with per-cell tri-quadratic geometry and eigen-decompositions (checked
    # against the numpy oracle in tests/test_oracle_cpu_port.py); its tables come from the numpy oracle, whose setup
    # loops over the cells in Python, hence the smaller sample mesh
    import asm_oracle_cpu

    threads = max(1, granted if args.cpu_threads <= 0 else min(usable, args.cpu_threads))
    nc = (24, 24, 24)
    mesh = orc.Mesh(nc, 4, [orc.DIRICHLET] * 3, mapfn=lambda pts: orc.kershaw(0.3, 0.3, pts))
    sm = asm_oracle_cpu.CpuSmootherDeformed(mesh, degree=args.degree, lambda_max=1.8, n_threads=threads, lib=asm_oracle_cpu.native_library())
    b = np.ones(mesh.n_dofs)
    x = sm.step(np.zeros(mesh.n_dofs), b, reps=2)  # warm-up
    per_call = 16
    reps, t0 = 0, time.perf_counter()
    while True:
        x = sm.step(x, b, reps=per_call)
        reps += per_call
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 4000:
            break
    return {
        "value": mesh.n_dofs * reps / el,
        "unit": "DoF/s",
        "cores": threads,
        "kind": "port",
        "host_cpu_count": os.cpu_count(),
        "host_cpus_usable": usable,
        "host_cpu_quota": quota,
        "sample": f"{reps} Chebyshev(k={args.degree})+FDM step() on {nc[0]}x{nc[1]}x{nc[2]} Kershaw cells p=4 "
        f"({mesh.n_dofs} DoFs), C++ restatement oracle/asm_oracle_cpu.cc (deformed-cell leg), {threads} threads",
    }


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout (file descriptor 1) when a communicator is created; stdout carries the
    one JSON line of this script and nothing else"""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cart_dirichlet", choices=["cart_dirichlet", "cart_periodic", "kershaw"])
    ap.add_argument("--cells", type=int, nargs=3, default=None, help="cells per GPU")
    ap.add_argument("--degree", type=int, default=1, help="Chebyshev degree (amd-opt_fdm1_cart.json: 1)")
    ap.add_argument("--what", default="cheb_step", choices=["cheb_step", "cheb_vmult"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "tiled"])
    ap.add_argument("--n-overlap", type=int, default=1, help="element-centred patches: cell + (n-1) DoF layers")
    ap.add_argument("--patch", default="cell", choices=["cell", "vertex"], help="element-centred or vertex-star patches")
    ap.add_argument("--weighting", default="post", choices=["none", "pre", "post", "symm", "ras"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra run at 128 x 128 x 96 cells (north_star_point)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = every core this process may use)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ASM_BENCH_SHARE_DEVICE"):  # rehearsal of the N>1 path on a one-GPU box: every rank on device 0
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    pkg = load_package()
    # the library (and with it the system's HIP runtime and RCCL it links) is loaded before torch, so that the
    # process has ONE HIP runtime and ONE RCCL, the ones the kernels were built against, whatever torch bundles
    # (ASM_BENCH_TORCH_FIRST=1: the other order - torch's bundled libamdhip64 / librccl then serve the library too)
    if not os.environ.get("ASM_BENCH_TORCH_FIRST"):
        pkg.load_library()
    dist = None
    if world > 1:
        # The process group only bootstraps the library's own RCCL communicator (broadcast of the
        # unique id) and carries the host-side barrier / max-over-ranks of the timing: gloo on CPU
        # tensors.  The data path (ghost planes) runs over RCCL inside the library; keeping torch's
        # NCCL backend out of the process means ONE RCCL and one HIP runtime are live in it.
        import torch
        import torch.distributed as dist

        dist.init_process_group(backend="gloo")

    default_workload = args.cells is None and args.workload == "cart_dirichlet" and args.degree == 1 and args.what == "cheb_step" \
        and args.n_overlap == 1 and args.patch == "cell" and args.weighting == "post" and args.kernel == "auto"
    cells = args.cells
    if cells is None:
        cells = {"cart_dirichlet": (80, 80, 80), "cart_periodic": (128, 64, 64), "kershaw": (78, 78, 78)}[args.workload]
    # weak scaling: the library splits z first, then y, then x
    probe = pkg.make_params(n_cells=(64, 64, 64), world_size=world, rank=rank)
    grid = list(pkg.describe_layout(probe).proc_grid)
    n_cells = [cells[d] * grid[d] for d in range(3)]
    boundary = "periodic" if args.workload == "cart_periodic" else "dirichlet"
    geometry = "kershaw" if args.workload == "kershaw" else "cartesian"
    extent = (float(n_cells[0]) / n_cells[2], float(n_cells[1]) / n_cells[2], 1.0) if geometry == "cartesian" else (1.0, 1.0, 1.0)
    ctx = pkg.Context(
        degree=4,
        n_cells=n_cells,
        boundary=boundary,
        geometry=geometry,
        kershaw_eps=(0.3, 0.3),
        extent=extent,
        precond="FDM",
        n_overlap=args.n_overlap,
        weighting=args.weighting,
        element_centric=args.patch == "cell",
        cheb_degree=args.degree,
        kernel=args.kernel,
        device=local_rank,
        rank=rank,
        world_size=world,
        proc_grid=grid,
    )
    if world > 1:
        import torch

        uid = [pkg.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        with stdout_to_stderr():
            ctx.comm_init(uid[0])
        # the timed loop modifies its vectors through the library only: the exchange of a step's result
        # (overlapped with the sweep that produces it) then serves the next step, as asm_set_keep_ghosts documents
        ctx.set_keep_ghosts(True)
    self_interface = world == 1 and os.environ.get("ASM_SELF_INTERFACE") and args.workload == "cart_periodic"
    if self_interface:
        # rehearsal of the multi-rank path on one GPU (tests/test_gpu_rccl_self.py): the rank exchanges its 26 ghost
        # messages with itself through RCCL; not a bench configuration, the line says so
        with stdout_to_stderr():
            ctx.comm_init(pkg.comm_unique_id())
        ctx.set_keep_ghosts(True)

    def barrier():
        # the library's two streams (sweeps, ghost exchange) carry all GPU work of this process:
        # asm_synchronize waits for both (what torch.cuda.synchronize() would do for torch's streams)
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    n_dofs = ctx.n_dofs
    b = ctx.vector(np.ones(ctx.n_owned))  # src = 1.0, matrix_free_loop_08.likwid.cc:304
    x = ctx.vector()
    if os.environ.get("ASM_BENCH_LAMBDA"):  # diagnostic builds whose results are wrong on purpose
        lam = float(os.environ["ASM_BENCH_LAMBDA"])
        ctx.set_eigenvalue(lam)
    else:
        lam = ctx.estimate_eigenvalues()  # setup, not timed (precondition.templates.h:461-463)
    run = {"cheb_step": ctx.cheb_step, "cheb_vmult": ctx.cheb_vmult}[args.what]
    for _ in range(args.warmup):
        run(x, b)
    barrier()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(x, b)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {k: ctx.profile_get(k) for k in ("laplace", "fdm")}
    ctx.profile_enable(False)
    if world > 1:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        a_sw, p_sw = algorithmic_bytes_per_dof(args.degree, args.what)
        # dominant kernel = the one with the larger total time
        kernels = {}
        for name, sweeps in (("laplace", a_sw), ("fdm", p_sw)):
            n, ms = prof[name]
            if n:
                bytes_per_launch = float(np.mean(sweeps)) * ctx.n_owned
                kernels[name] = {
                    "launches": n,
                    "avg_ms": ms / n,
                    "bytes_per_launch": bytes_per_launch,
                    "GBps": bytes_per_launch / (ms / n * 1e-3) / 1e9,
                }
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
        # HBM traffic per launch: measured separately with rocprofv3 PMC passes (bench.py cannot
        # run the profiler on itself); only reported when the committed measurement is for
        # exactly this workload and kernel family, else null
        traffic, traffic_src = None, None
        key = f"{args.workload}:{cells[0]}x{cells[1]}x{cells[2]}:{ctx.kernel_variant()}:{args.what}:{args.degree}"
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        except OSError:
            tr = {}
        pair_a, pair_p = ctx.pair_sweeps()
        lap_variant = "generic" if args.workload == "kershaw" else ("wave-pair" if pair_a else ctx.kernel_variant())
        fdm_variant = "wave-pair" if pair_p else ctx.kernel_variant()
        # (entries per kernel family: the context picks the family per sweep)
        dom_key = f"{dom}:{lap_variant if dom == 'laplace' else fdm_variant}"
        if world == 1 and key in tr and dom_key in tr[key]:
            traffic, traffic_src = tr[key][dom_key], tr[key]["source"]
        # the deformed-cell Laplace sweep is bound by fp64 issue, not by memory (report the FLOP rate then)
        fp64_bound = args.workload == "kershaw" and dom == "laplace"
        if fp64_bound:
            flops = KERSHAW_LAPLACE_FLOP_PER_DOF * ctx.n_owned
            kernels["laplace"]["TFLOPps"] = flops / (kernels["laplace"]["avg_ms"] * 1e-3) / 1e12
        roofline = {
            "bound": "fp64" if fp64_bound else "hbm",
            "kernel": {"laplace": f"Laplace sweep ({lap_variant})", "fdm": f"FDM Schwarz sweep ({fdm_variant})"}[dom],
            "achieved": kernels[dom]["GBps"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_launch_ms": kernels[dom]["avg_ms"],
            "kernels": kernels,
        }
        if fp64_bound:
            roofline.update({"achieved": kernels["laplace"]["TFLOPps"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": kernels["laplace"]["TFLOPps"] / FP64_PEAK_TFLOPS,
                             "hbm_frac": kernels["laplace"]["GBps"] / HBM_PEAK_GBS,
                             "flop_per_dof": KERSHAW_LAPLACE_FLOP_PER_DOF})
        for k in kernels.values():
            k["hbm_frac"] = k["GBps"] / HBM_PEAK_GBS
        total_bytes = (sum(a_sw) + sum(p_sw)) * n_dofs
        out = {
            "metric": "DoF/s per smoother application (Cheby+FDM, p=4, 3D)",
            "value": n_dofs * args.steps / elapsed,
            "unit": "DoF/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload} p=4 {n_cells[0]}x{n_cells[1]}x{n_cells[2]} cells, "
                f"Chebyshev degree {args.degree} {args.what.split('_')[1]}(), FDM "
                + (f"n_overlap={args.n_overlap}" if args.patch == "cell" else "vertex-star")
                + f" {args.weighting} weights",
                "n_dofs": n_dofs,
                "cells_per_gpu": list(cells),
                "proc_grid": grid,
                "kernel_variant": ctx.kernel_variant(),
                "laplace_rows": ctx.tiled_laplace_rows(),
                "pair_sweeps": {"laplace": pair_a, "fdm": pair_p, "z_chunks": ctx.pair_zchunks(), "waves": ctx.pair_waves()},
                "lambda_max_estimate": lam,
                **({"rehearsal": "periodic directions closed by RCCL messages of the rank to itself"} if self_interface else {}),
            },
            "application_bytes_per_dof": (sum(a_sw) + sum(p_sw)),
            "application_hbm_frac": total_bytes / world * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
            "roofline": roofline,
        }
        if world == 1:
            # what a plain copy sustains on this device (read + written bytes; 1 GiB arrays, far beyond the 256 MiB Infinity
            # Cache): the achievable HBM rate next to the 8 TB/s peak every fraction above is quoted against
            b.free()
            x.free()
            out["peak_measured"] = {"copy_GBps": ctx.measure_copy_bandwidth(1 << 30, 5), "unit": "GB/s",
                                    "kernel": "16-byte non-temporal copy, 4 loads in flight per lane, 2 x 1 GiB"}
        if world == 1 and default_workload and not args.no_north_star:
            # the size the north star is quoted on (p = 4, ~1e8 DoFs), same step, same accounting; the headline `value`
            # stays on BASELINE config 2 (~3e7 DoFs) so that rounds compare
            ctx.close()
            ns = pkg.Context(degree=4, n_cells=NORTH_STAR_CELLS, boundary="dirichlet", geometry="cartesian",
                             extent=(NORTH_STAR_CELLS[0] / NORTH_STAR_CELLS[2], NORTH_STAR_CELLS[1] / NORTH_STAR_CELLS[2], 1.0),
                             precond="FDM", n_overlap=1, weighting="post", cheb_degree=1, device=local_rank)
            nb, nx = ns.vector(np.ones(ns.n_owned)), ns.vector()
            ns.set_eigenvalue(lam)
            for _ in range(3):
                ns.cheb_step(nx, nb)
            ns.synchronize()
            ns.profile_enable(True)
            t1 = time.perf_counter()
            ns_steps = 10
            for _ in range(ns_steps):
                ns.cheb_step(nx, nb)
            ns.synchronize()
            el = time.perf_counter() - t1
            nk = {}
            for name in ("laplace", "fdm"):
                n, ms = ns.profile_get(name)
                nk[name] = {"avg_ms": ms / n, "GBps": 24.0 * ns.n_owned / (ms / n * 1e-3) / 1e9,
                            "hbm_frac": 24.0 * ns.n_owned / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["north_star_point"] = {"cells": list(NORTH_STAR_CELLS), "n_dofs": ns.n_dofs, "steps": ns_steps,
                                       "ms_per_step": el / ns_steps * 1e3, "value": ns.n_dofs * ns_steps / el, "unit": "DoF/s",
                                       "laplace_rows": ns.tiled_laplace_rows(),
                                       "pair_sweeps": dict(zip(("laplace", "fdm", "z_chunks", "waves"), ns.pair_sweeps() + (ns.pair_zchunks(), ns.pair_waves()))), "kernels": nk}
            ns.close()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, pkg, cells)
        # the line the reference harness prints for the same run (matrix_free_loop_08.likwid.cc:390-395:
        # ">> label n_dofs reps time sizeof(Number) degree n_ghost n_import", label DSL :108-144, 249-298:
        # cheby-<degree>-<optimize>-<weight>-<n_overlap | v>-<weight sequence>); on stderr: stdout carries the JSON line only
        label = (f"cheby-{args.degree}-2-{'add' if args.weighting == 'none' else args.weighting}-"
                 f"{'v' if args.patch == 'vertex' else args.n_overlap}-c")
        print(f">> {label} {n_dofs} {args.steps * args.degree} {elapsed:.6e} 8 4 {int(ctx.layout.n_ghost)} {int(ctx.layout.n_ghost)}",
              file=sys.stderr)
        print(json.dumps(out))
    ctx.close()  # (idempotent)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
