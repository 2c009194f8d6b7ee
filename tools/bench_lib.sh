#!/bin/bash
# bench.py with another build of the library: bench_lib.sh <path to .so> [bench args]
export DEALII_ASM_AMD_LIB="$1"; shift
python "$(dirname "$0")/../bench.py" "$@" 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
n = d.get('north_star_point')
print(round(d['ms_per_step'], 4), {k: round(v['avg_ms'], 4) for k, v in d['roofline']['kernels'].items()}, d['config'].get('pair_sweeps'), '| north star:', (round(n['ms_per_step'], 4), {k: round(v['avg_ms'], 4) for k, v in n['kernels'].items()}, n.get('pair_sweeps')) if n else None)"
