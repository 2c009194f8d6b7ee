"""Build-time audit of the wave-pair kernels (dealii-asm_amd/csrc/kernels_pair.hip).  They keep the next layer's source
planes and update operands in FIXED accumulation registers (two waves per tile: a92..a255, a28..a255 for the three-term
update; four waves per tile: a18..a127, and 256 registers per wave in all) which the compiler does not know about.  That is sound only while the compiled code uses no scratch memory, touches no
accumulation register from that boundary upwards and contains no flat_ instruction (whose completion the kernels' wait
counts do not cover): tools/audit_pair.sh compiles the file for gfx950 and checks the ISA.  No GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_pair_kernels_pass_the_register_audit():
    res = subprocess.run([os.path.join(ROOT, "tools", "audit_pair.sh")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("ok")]
    assert len(lines) == 11, res.stdout  # one per fused-update mode with two waves per tile, five with four
