"""ctypes driver of oracle/asm_oracle_cpu.cc (TEST INFRASTRUCTURE ONLY: tests/ and the
cpu_baseline leg of bench.py).  Tables come from the numpy oracle."""
import ctypes as C
import os

import numpy as np

import asm_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle_cpu.so")


def spread_cpus(n):
    """n CPUs for the pool's threads (ASM_ORACLE_CPUS): one hardware thread per physical core, spread evenly over the cores
    of ONE socket this process may use - the threads then share no core with an SMT sibling, have the socket's L3 slices
    between them and meet at their barriers inside one NUMA node (measured on the 2 x 64-core host of the GPU boxes with 16
    threads: 4.7e8 DoF/s against 3.8e8 unpinned and 4.1e8 spread over both sockets, profiles/r03_cpu_baseline_threads.txt).
    None if the topology cannot be read or the socket has fewer cores than threads."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        sockets = {}
        for c in allowed:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "thread_siblings_list") as f:
                first = int(f.read().replace("-", ",").split(",")[0])
            if first == c or first not in allowed:
                with open(base + "physical_package_id") as f:
                    sockets.setdefault(int(f.read()), []).append(c)
    except (OSError, ValueError, AttributeError):
        return None
    if not sockets:
        return None
    cores = max(sockets.values(), key=len)
    if n < 1 or len(cores) < n:
        return None
    return [cores[(i * len(cores)) // n] for i in range(n)]


def native_library():
    """the C++ restatement built with -march=native ON THIS MACHINE (bench.py's cpu_baseline), or the
    portable build if that fails.  The file name carries a hash of the CPU's model and flags: a build made on another
    machine (the snapshot travels) is not mistaken for this one's."""
    import hashlib
    import subprocess

    try:
        info = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
        tag = hashlib.sha1("".join(info).encode()).hexdigest()[:12]
        subprocess.run(["make", "-C", HERE, "native", f"NATIVE_TAG={tag}"], check=True, capture_output=True, timeout=300)
        path = os.path.join(HERE, "_build", f"liboracle_cpu_native_{tag}.so")
        C.CDLL(path)
        return path
    except Exception:  # noqa: BLE001 - any failure: the portable build
        return LIB


def _ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class CpuSmoother:
    """Chebyshev(k) + FDM (n_overlap=1, post weights) on a Cartesian box, degree 4."""

    def __init__(self, nc, periodic, extent=(1.0, 1.0, 1.0), degree=3, lambda_max=1.8, smoothing_range=20.0, n_threads=1, lib=None):
        if not os.path.exists(lib or LIB):
            raise RuntimeError(f"{lib or LIB} missing: run `make -C oracle`")
        self.lib = C.CDLL(lib or LIB)
        self.nc = np.array(nc, dtype=np.int32)
        self.periodic = np.array([1 if periodic else 0] * 3, dtype=np.int32)
        self.h = np.array([extent[d] / nc[d] for d in range(3)], dtype=np.float64)
        self.degree, self.n_threads = degree, n_threads
        max_ev = 1.2 * lambda_max
        min_ev = max_ev / smoothing_range
        self.theta, self.delta = 0.5 * (max_ev + min_ev), 0.5 * (max_ev - min_ev)
        b = orc.Basis1D(4)
        self.N = np.ascontiguousarray(b.N)
        _, Dc = orc.lagrange_eval(b.xq, b.xq)
        self.Dc = np.ascontiguousarray(Dc)
        self.wq = np.ascontiguousarray(b.wq)
        # 1-D FDM instances and weights per direction (tensor-product structure of a box)
        eig, self.inst, self.wpost = [], [], []
        cache = {}
        for d in range(3):
            inst = np.zeros(nc[d], dtype=np.int32)
            w = np.zeros((nc[d], 5))
            for c in range(nc[d]):
                lo = "internal" if (periodic or c > 0) else orc.DIRICHLET
                hi = "internal" if (periodic or c < nc[d] - 1) else orc.DIRICHLET
                h3 = (self.h[d] if lo == "internal" else 0.0, self.h[d], self.h[d] if hi == "internal" else 0.0)
                key = (h3, lo, hi)
                if key not in cache:
                    M, K = orc.fdm_matrices_element(b, h3, lo, hi, 1)
                    lam, Q = orc.fdm_eigen(M, K)
                    cache[key] = len(eig)
                    eig.append(np.concatenate([Q.reshape(-1), lam]))
                inst[c] = cache[key]
                for i in range(5):
                    valence = 2.0 if (i in (0, 4)) else 1.0
                    if i == 0 and lo != "internal":
                        valence = 0.0
                    if i == 4 and hi != "internal":
                        valence = 0.0
                    w[c, i] = 0.0 if valence == 0 else 1.0 / valence
            self.inst.append(inst)
            self.wpost.append(np.ascontiguousarray(w))
        self.eig = np.ascontiguousarray(np.array(eig))
        self.n_dofs = int(np.prod([nc[d] * 4 + (0 if periodic else 1) for d in range(3)]))

    def step(self, x, b, reps=1):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        b = np.ascontiguousarray(b, dtype=np.float64)
        err = self.lib.oracle_cpu_cheb_step(
            _ptr(self.nc, C.c_int32), _ptr(self.periodic, C.c_int32), _ptr(self.h), _ptr(self.N), _ptr(self.Dc), _ptr(self.wq),
            _ptr(self.inst[0], C.c_int32), _ptr(self.inst[1], C.c_int32), _ptr(self.inst[2], C.c_int32), _ptr(self.eig),
            _ptr(self.wpost[0]), _ptr(self.wpost[1]), _ptr(self.wpost[2]),
            C.c_int(self.degree), C.c_double(self.theta), C.c_double(self.delta), _ptr(x), _ptr(b), C.c_int(self.n_threads), C.c_int(reps))
        if err:
            raise RuntimeError("oracle_cpu_cheb_step failed (odd number of cells in a periodic direction?)")
        return x


class CpuSmootherDeformed:
    """Chebyshev(k) + FDM (n_overlap=1, post weights) on a deformed Dirichlet mesh (tri-quadratic geometry), degree 4:
    per-cell map coefficients, eigen-decompositions and patch weights come from the numpy oracle's objects (all of
    them array operations, the eigen-decompositions batched for large meshes: the 78^3-cell mesh of the benchmarks is
    within reach)."""

    def __init__(self, mesh, degree=3, lambda_max=1.8, smoothing_range=20.0, n_threads=1, lib=None):
        if not os.path.exists(lib or LIB):
            raise RuntimeError(f"{lib or LIB} missing: run `make -C oracle`")
        assert mesh.p == 4 and mesh.dim == 3 and not any(mesh.periodic)
        self.lib = C.CDLL(lib or LIB)
        self.nc = np.array(mesh.nc, dtype=np.int32)
        self.degree, self.n_threads, self.n_dofs = degree, n_threads, mesh.n_dofs
        max_ev = 1.2 * lambda_max
        min_ev = max_ev / smoothing_range
        self.theta, self.delta = 0.5 * (max_ev + min_ev), 0.5 * (max_ev - min_ev)
        b = mesh.basis
        self.N = np.ascontiguousarray(b.N)
        _, Dc = orc.lagrange_eval(b.xq, b.xq)
        self.Dc, self.wq, self.xq = np.ascontiguousarray(Dc), np.ascontiguousarray(b.wq), np.ascontiguousarray(b.xq)
        # (the operator object itself - Jacobians at every quadrature point of every cell - is not needed: the C++ leg
        # evaluates them from the 27 coefficient vectors, as the reference does)
        prec = orc.ASPoissonPreconditioner(mesh, 1, "post")
        self.q2 = np.ascontiguousarray(orc.LaplaceOperator.q2_monomial_coefficients(mesh).reshape(mesh.ncells, 81))  # [cell][9k+3j+i][e]
        eig = np.concatenate([prec.Q.reshape(mesh.ncells, 3, 25), prec.lam.reshape(mesh.ncells, 3, 5)], axis=2)
        self.eig = np.ascontiguousarray(eig)  # [cell][d][Q(i, a) row-major | lambda]
        self.wpost = np.ascontiguousarray(prec.w_post.reshape(mesh.ncells, 125))

    def step(self, x, b, reps=1):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        b = np.ascontiguousarray(b, dtype=np.float64)
        err = self.lib.oracle_cpu_cheb_step_deformed(
            _ptr(self.nc, C.c_int32), _ptr(self.N), _ptr(self.Dc), _ptr(self.wq), _ptr(self.xq), _ptr(self.q2), _ptr(self.eig),
            _ptr(self.wpost), C.c_int(self.degree), C.c_double(self.theta), C.c_double(self.delta), _ptr(x), _ptr(b),
            C.c_int(self.n_threads), C.c_int(reps))
        if err:
            raise RuntimeError("oracle_cpu_cheb_step_deformed failed")
        return x
