"""Diagnostic: life of every workgroup of the tiled sweeps (start / end in s_memrealtime ticks of 10 ns, compute unit).
Needs a library built with -DASM_STAMP (tools/ablate.sh name:-DASM_STAMP ...): python tools/wgtrace_run.py name"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DEALII_ASM_AMD_LIB"] = os.path.join(ROOT, "dealii-asm_amd", "abl", "lib_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamp"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
cells = tuple(int(x) for x in os.environ.get("STAMP_CELLS", "80,80,80").split(","))
ctx = pkg.Context(degree=4, n_cells=cells, boundary="dirichlet", cheb_degree=1)
ctx.set_eigenvalue(1.7)
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lib = ctx.lib
lib.asm_debug_wgtrace.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.asm_debug_wgtrace.restype = C.c_int
for _ in range(5):
    ctx.cheb_step(x, b)
lib.asm_debug_wgtrace(None, 0, 1)
ctx.cheb_step(x, b)
buf = np.zeros((16384, 4), dtype=np.uint64)
n = lib.asm_debug_wgtrace(buf.ctypes.data_as(C.c_void_p), 16384, 1)
t = buf[:n]
for tag, name in ((0, "Laplace"), (1, "FDM")):
    sel = t[(t[:, 3] >> np.uint64(32)) == tag]
    if len(sel) == 0:
        continue
    t0 = sel[:, 0].min()
    st, en = (sel[:, 0] - t0).astype(float) / 100.0, (sel[:, 1] - t0).astype(float) / 100.0  # microseconds
    hw = sel[:, 2] & np.uint64(0xFFFFFFFF)
    xcc = sel[:, 2] >> np.uint64(32)
    cu = (xcc.astype(np.int64) & 0xF) * 4096 + (hw.astype(np.int64) >> 8 & 0xF) + 16 * (hw.astype(np.int64) >> 13 & 0x7)  # xcc, cu_id, se_id
    dur = en - st
    print(f"{name}: {len(sel)} workgroups, kernel span {en.max():.1f} us; workgroup life: mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us")
    print(f"  first start spread {st.min():.1f}..{np.sort(st)[min(255, len(st) - 1)]:.1f} us; distinct CUs seen: {len(set(cu.tolist()))}")
    ends = {}
    gaps, per_cu_end, per_cu_busy = [], [], []
    for c in set(cu.tolist()):
        m = cu == c
        o = np.argsort(st[m])
        s_, e_ = st[m][o], en[m][o]
        per_cu_end.append(e_.max())
        per_cu_busy.append((e_ - s_).sum())
        gaps += list(s_[1:] - e_[:-1])
    gaps = np.array(gaps)
    print(f"  per CU: last end mean {np.mean(per_cu_end):.1f} min {np.min(per_cu_end):.1f} max {np.max(per_cu_end):.1f} us; busy mean {np.mean(per_cu_busy):.1f} us")
    if len(gaps):
        print(f"  gap between consecutive workgroups on a CU: mean {gaps.mean():.2f} median {np.median(gaps):.2f} max {gaps.max():.2f} us (n={len(gaps)})")
    h, edges = np.histogram(dur, bins=8)
    print("  life histogram (us):", ", ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}: {h[i]}" for i in range(len(h))))
    # life by cost class (needs ASM_NO_TILED_ORDER=1: launch position = work index)
    if os.environ.get("ASM_NO_TILED_ORDER"):
        nb = len(sel)
        b = (sel[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        xcd, idx = b & 7, b >> 3
        i = xcd * (nb >> 3) + np.minimum(xcd, nb & 7) + idx
        ntx, nty = (cells[0] + 10) // 11, (cells[1] + 6) // 7
        nzc = nb // (ntx * nty)
        tx, ty, zc = i % ntx, (i // ntx) % nty, i // (ntx * nty)
        for name2, m in (("interior", (tx > 0) & (tx < ntx - 1) & (ty > 0) & (ty < nty - 1) & (zc > 0) & (zc < nzc - 1)),
                         ("x low only", (tx == 0) & (ty > 0) & (ty < nty - 1) & (zc > 0) & (zc < nzc - 1)),
                         ("x high only", (tx == ntx - 1) & (ty > 0) & (ty < nty - 1) & (zc > 0) & (zc < nzc - 1)),
                         ("y low only", (ty == 0) & (tx > 0) & (tx < ntx - 1) & (zc > 0) & (zc < nzc - 1)),
                         ("y high only", (ty == nty - 1) & (tx > 0) & (tx < ntx - 1) & (zc > 0) & (zc < nzc - 1)),
                         ("z low only", (zc == 0) & (tx > 0) & (tx < ntx - 1) & (ty > 0) & (ty < nty - 1)),
                         ("z high only", (zc == nzc - 1) & (tx > 0) & (tx < ntx - 1) & (ty > 0) & (ty < nty - 1))):
            if m.any():
                print(f"  class {name2:12s}: n={m.sum():4d} life mean {dur[m].mean():.1f} us (min {dur[m].min():.1f}, max {dur[m].max():.1f})")
