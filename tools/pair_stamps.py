"""Diagnostic: cycles (s_memtime) one interior wave of the wave-pair kernels spends per stage and layer.
Needs tools/pair_stamps.sh (library built with -DASM_PAIR_STAMP).  usage: pair_stamps.py [nx ny nz]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DEALII_ASM_AMD_LIB"] = os.path.join(ROOT, "dealii-asm_amd", "abl", "lib_pairstamp.so")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

nc = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (80, 80, 80)
pkg = load_package()
ctx = pkg.Context(degree=4, n_cells=nc, boundary="dirichlet", cheb_degree=1)
ctx.set_eigenvalue(1.7)
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lib = ctx.lib
lib.asm_debug_pair_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
for _ in range(3):
    ctx.cheb_step(x, b)
lib.asm_debug_pair_stamps(out, 1)
n = 10
for _ in range(n):
    ctx.cheb_step(x, b)
lib.asm_debug_pair_stamps(out, 1)
names = ["emit tail + operand requests -> top of layer", "diagonal refresh, wait, forward x-y of two planes, plane requests", "barrier 1",
         "z columns", "barrier 2", "(after the loop)"]
layers = out[8]
print("cells", nc, "- A and P sweeps together, %d wave-layers" % layers)
tot = sum(out[i] for i in range(5))
for i in range(5):
    print("  %-70s %8.0f cycles/layer  %5.1f %%" % (names[i], out[i] / layers, 100.0 * out[i] / tot))
print("  %-70s %8.0f cycles/layer" % ("total (stage 0 holds backward x-y, merge, update, stores, barrier 3)", tot / layers))
