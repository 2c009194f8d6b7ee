"""Diagnostic: per-phase cycles (s_memtime) of wave 0 / workgroup 0 of the tiled FDM kernel.
Needs a library built with -DASM_STAMP (dealii-asm_amd/abl/lib_stamp.so)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DEALII_ASM_AMD_LIB"] = os.path.join(ROOT, "dealii-asm_amd", "abl", "lib_stamp.so")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
ctx = pkg.Context(degree=4, n_cells=(80, 80, 80), boundary="dirichlet", cheb_degree=1)
ctx.set_eigenvalue(1.7)
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lib = ctx.lib
lib.asm_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
for _ in range(3):
    ctx.cheb_step(x, b)
lib.asm_debug_stamps(ctx.handle, out, 1)
n = 10
for _ in range(n):
    ctx.cheb_step(x, b)
lib.asm_debug_stamps(ctx.handle, out, 1)
names = ["loop top", "x,y fwd (+gather wait)", "transpose A->B", "z sweeps+scale", "transpose B->A", "y,x bwd + weights", "(unused)", "prefetch ops + carry", "acc even + gather issue", "barrier 1", "acc odd + gather issue", "barrier 2", "store epilogue", "barrier 3", "", ""]
tot = sum(out)
layers = 17 * n
for i in range(14):
    print(f"{names[i]:26s} {out[i] / layers:10.0f} cycles/layer  {100.0 * out[i] / tot:5.1f}%")
print("total per layer", tot / layers)
