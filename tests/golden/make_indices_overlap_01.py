"""Turns the reference's own fixture indices_overlap_01.output (DoF index lists of element-centred
patches, 2-D, FE_Q(2), hyper_cube refined 3 times, n_overlap = 0..3; produced by
indices_overlap_01.cc through DoFTools::get_dof_indices_cell_with_overlap, include/dof_tools.h:10-137)
into tests/golden/indices_overlap_01.json.  Data only: numbers in, numbers out."""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/indices_overlap_01.output"
cells, cur = [], []
for line in open(src):
    t = line.split()
    if not t:
        if cur:
            cells.append(cur)
        cur = []
    else:
        cur.append([int(x) for x in t])
if cur:
    cells.append(cur)
assert len(cells) == 64 and all(len(c) == 4 for c in cells)
out = {
    "_source": "indices_overlap_01.output (dim 2, fe_degree 2, 3 global refinements; cells in deal.II order, n_overlap 0..3)",
    "degree": 2,
    "level": 3,
    "cells": cells,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "indices_overlap_01.json"), "w"))
