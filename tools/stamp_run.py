"""Diagnostic: per-phase cycles (s_memtime) of one interior wave of the tiled kernels.
Needs a library built with -DASM_STAMP (dealii-asm_amd/abl/lib_stamp.so)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DEALII_ASM_AMD_LIB"] = os.path.join(ROOT, "dealii-asm_amd", "abl", "lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamp"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
ctx = pkg.Context(degree=4, n_cells=(80, 80, 80), boundary="dirichlet", cheb_degree=1)
ctx.set_eigenvalue(1.7)
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lib = ctx.lib
lib.asm_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 32)()
for _ in range(3):
    ctx.cheb_step(x, b)
lib.asm_debug_stamps(ctx.handle, out, 1)
n = 10
for _ in range(n):
    ctx.cheb_step(x, b)
lib.asm_debug_stamps(ctx.handle, out, 1)
names_l = ["(loop overhead)", "LDS image read", "x + z fwd sweeps, DMA requests", "transpose", "y sweeps, scaling, operand requests", "transpose",
           "z + x bwd sweeps", "merge x/z (lane shift)", "fused update, stores issue", "", "y slot write", "barrier", "y slot read",
           "wait: staged rows + operands"]
names_f = ["(loop overhead)", "LDS image read, tables, weights", "x + z fwd sweeps", "transpose", "y sweeps, scaling, DMA + operand requests",
           "transpose", "z + x bwd sweeps, weights", "merge x/z (lane shift)", "fused update, stores issue", "", "y slot write", "barrier",
           "y slot read", "wait: staged rows + operands"]
layers = 11 * n  # 8 z chunks of 10 layers + halo layer
for title, base, names in (("Laplace (OP_RESIDUAL)", 0, names_l), ("FDM (PC_STEP1)", 16, names_f)):
    tot = sum(out[base + i] for i in range(16))
    print(title)
    for i in range(14):
        if names[i]:
            print(f"  {names[i]:34s} {out[base + i] / layers:10.0f} cycles/layer  {100.0 * out[base + i] / max(tot, 1):5.1f}%")
    print(f"  total per layer {tot / layers:.0f} cycles (s_memtime ticks)")
