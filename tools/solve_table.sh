#!/bin/bash
# usage (GPU box): tools/solve_table.sh -> the whole-solve rows of DESIGN.md 5.4 (profiles/r02_whole_solve_timings.txt)
for coarse in cg direct; do
  python tools/solve_run.py --cells 48 48 48 --coarse $coarse
  python tools/solve_run.py --cells 80 80 80 --coarse $coarse
  python tools/solve_run.py --cells 48 48 48 --geometry kershaw --degree 2 --coarse $coarse
done
python tools/solve_run.py --cells 48 48 48 --geometry kershaw --degree 2 --n-overlap 2 --coarse direct
python tools/solve_run.py --cells 48 48 48 --geometry kershaw --degree 2 --solver CG --weighting symm --patch vertex --coarse direct
