"""Transcribes '- n iterations:' and the options that define each run from the reference's regression fixtures
tests/element_centered_preconitioners/small/*.{json,output} into tests/golden/reference_small_iterations.json.
Data only (numbers and option strings)."""
import json
import os
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/element_centered_preconitioners/small"
out = {"_comment": "2-D, degree 3, hyper_cube refined 3 times (8x8 cells, 625 DoFs), rhs f=1, GMRES (right preconditioning, "
       "classical Gram-Schmidt), ReductionControl(1000, 1e-10, 1e-2); 'it' = SolverControl::last_step()", "runs": []}
for name in sorted(os.listdir(src)):
    if not name.endswith(".json"):
        continue
    cfg = json.load(open(os.path.join(src, name)))
    txt = open(os.path.join(src, name.replace(".json", ".output"))).read()
    m = re.search(r"n iterations:\s+(\d+)", txt)
    out["runs"].append({"file": name.replace(".json", ".output"), "solver": cfg["solver"]["type"],
                        "preconditioner": cfg["preconditioner"], "it": int(m.group(1))})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_small_iterations.json"), "w"), indent=1)
