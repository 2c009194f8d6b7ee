"""Rehearsal on a one-GPU box of what bench.py --gpus N does around the library when N > 1:
the library (linked against /opt/rocm's RCCL and HIP runtime) and torch.distributed's NCCL backend
(torch bundles its own librccl.so / libamdhip64.so with the same sonames) in one process, in the
import order bench.py uses.  World size 1, so RCCL never needs a second device."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

pkg = load_package()
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
uid = [pkg.comm_unique_id()]
dist.broadcast_object_list(uid, src=0)
ctx = pkg.Context(degree=4, n_cells=(8, 8, 8), boundary="dirichlet", cheb_degree=1, device=0, rank=0, world_size=1)
ctx.comm_init(uid[0])
b = ctx.vector(np.ones(ctx.n_owned))
x = ctx.vector()
lam = ctx.estimate_eigenvalues()
ctx.cheb_step(x, b)
ctx.synchronize()
t = torch.tensor([1.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "librccl" in l or "libamdhip64" in l})
print("lambda", lam, "norm", float(np.linalg.norm(x.get())), "allreduce", float(t.item()))
print("loaded:", libs)
dist.destroy_process_group()
