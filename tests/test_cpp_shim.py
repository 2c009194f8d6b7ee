"""The C++ shim (include/dealii_asm_amd.hpp) compiles with the host compiler alone,
links against the in-tree library and behaves: refuses on CPU, smooths on a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dealii-asm_amd")


def _build(tmp_path, name="shim_smoke"):
    exe = str(tmp_path / name)
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cc"),
         "-L", LIBDIR, "-ldealii_asm_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
        check=True,
    )
    return exe


def test_shim_compiles_and_refuses_without_gpu(pkg, tmp_path):
    pkg.load_library()
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_runs_smoother_on_gpu(pkg, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "min ev" in r.stdout


def test_fused_driver_compiles_and_refuses_without_gpu(pkg, tmp_path):
    pkg.load_library()
    exe = _build(tmp_path, "fused_chebyshev")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr


@pytest.mark.gpu
def test_fused_driver_matches_step_bitwise_on_gpu(pkg, tmp_path):
    """a hand-written Chebyshev loop over vmult(dst, src, FusedUpdate) == PreconditionChebyshev::step, bit for bit"""
    exe = _build(tmp_path, "fused_chebyshev")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bitwise identical" in r.stdout


def test_multigrid_driver_compiles_and_refuses_without_gpu(pkg, tmp_path):
    pkg.load_library()
    exe = _build(tmp_path, "multigrid_solve")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr


@pytest.mark.gpu
def test_multigrid_driver_converges_on_gpu(pkg, tmp_path):
    exe = _build(tmp_path, "multigrid_solve")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GMRES it" in r.stdout
