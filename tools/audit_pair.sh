#!/bin/bash
# Audit of the wave-pair kernels (dealii-asm_amd/csrc/kernels_pair.hip): they keep the next layer's planes and operands in
# FIXED accumulation registers a28..a255 the compiler does not know about.  That is only sound while the compiler itself
# (a) uses no scratch memory and (b) touches no accumulation register from a28 (three-term update) / a92 (other modes, two
# waves per tile) / a18 (four waves per tile, which also must fit 256 registers) upwards, (c) emits no flat_ instruction.  Run after every edit of the file.
set -e
cd "$(dirname "$0")/../dealii-asm_amd"
out=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I/opt/rocm/include --cuda-device-only -S -o $out/pair.s csrc/kernels_pair.hip 2>/dev/null
python3 - $out/pair.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN6asmamd\S*pair_sweep_kernel\S*):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    inasm, worst, n = False, -1, 0
    for line in body.split('\n'):
        if 'ASMSTART' in line: inasm = True
        elif 'ASMEND' in line: inasm = False
        elif not inasm and 'v_accvgpr' in line:
            n += 1
            for r in re.findall(r'\ba(\d+)\b', line):
                worst = max(worst, int(r))
            for r in re.findall(r'a\[(\d+):(\d+)\]', line):
                worst = max(worst, int(r[1]))
    def setval(key):
        mm = re.search(r'\.set ' + re.escape(name) + r'\.' + key + r', (\d+)', txt)
        return int(mm.group(1)) if mm else -1
    scratch = setval('private_seg_size')
    vg = '%d+%d' % (setval('num_vgpr'), setval('num_agpr'))
    mode, waves = (int(x) for x in re.search(r'pair_sweep_kernelILi(\d)ELi(\d)', name).groups())
    limit = 18 if waves == 4 else (28 if mode == 5 else 92)  # ARegs<NEED_B, NW>::BASE
    regs = setval('num_vgpr') + setval('num_agpr')
    if waves == 4 and regs > 256:  # two waves per SIMD
        limit = -1
    ok = scratch == 0 and worst < limit and 'flat_' not in body
    bad += 0 if ok else 1
    print('%-4s %s: scratch %d B, vgpr+agpr %s, compiler accvgpr ops %d (highest a%d)' % ('ok' if ok else 'FAIL', re.search(r'pair_sweep_kernelILi(\d)ELi(\d)', name).group(0), scratch, vg, n, worst))
sys.exit(1 if bad else 0)
PY
rm -rf $out
