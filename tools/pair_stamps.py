"""Diagnostic: cycles (s_memtime) one interior wave of the wave-pair kernels spends per stage and layer.
Needs tools/pair_stamps.sh (library built with -DASM_PAIR_STAMP).  usage: pair_stamps.py [nx ny nz]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DEALII_ASM_AMD_LIB"] = os.path.join(ROOT, "dealii-asm_amd", "abl", os.environ.get("PAIR_STAMP_LIB", "lib_pairstamp.so"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

nc = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (80, 80, 80)
bc = sys.argv[4] if len(sys.argv) >= 5 else "dirichlet"
pkg = load_package()
ctx = pkg.Context(degree=4, n_cells=nc, boundary=bc, cheb_degree=1)
ctx.set_eigenvalue(1.7)
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lib = ctx.lib
lib.asm_debug_pair_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * (6 * 24))()
for _ in range(3):
    ctx.cheb_step(x, b)
lib.asm_debug_pair_stamps(out, 1)
n = 10
for _ in range(n):
    ctx.cheb_step(x, b)
lib.asm_debug_pair_stamps(out, 1)
names = {0: "operand requests of the next layer -> top of layer", 6: "z class, diagonal refresh", 7: "wait for the planes", 8: "unpack, forward x-y of two planes, LDS writes",
         1: "plane requests of the next layer", 2: "barrier 1", 3: "z columns", 4: "barrier 2", 9: "LDS reads of output plane 0", 10: "wait for the operands",
         11: "backward x-y, merge, update, store of plane 0", 12: "LDS reads of output plane 1", 13: "barrier 3", 14: "backward x-y, merge, update, store of plane 1"}
order = [6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 14, 0]
for mode, title in ((1, "Laplace sweep, residual form"), (4, "FDM sweep, two-term update")):
    o = out[24 * mode:24 * mode + 24]
    layers = o[16]
    if layers == 0:
        print(title, ": the context did not take the wave-pair kernel (ASM_PAIR=1 forces it)")
        continue
    print("cells", nc, bc, "-", title, "- %d wave-layers" % layers)
    tot = sum(o[i] for i in order)
    for i in order:
        print("  %-70s %8.0f cycles/layer  %5.1f %%" % (names[i], o[i] / layers, 100.0 * o[i] / tot))
    print("  %-70s %8.0f cycles/layer" % ("total", tot / layers))
