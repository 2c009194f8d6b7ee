"""Regenerates reference_small_outputs.json from the reference's own fixtures
(/root/reference/tests/element_centered_preconitioners/small/*.output and
subdivided_hyper_cube_balanced_01.output).  Only NUMBERS are extracted (data the
reference's tests hold); run it where /root/reference exists."""
import json
import os
import re

REF = "/root/reference"
SMALL = os.path.join(REF, "tests/element_centered_preconitioners/small")


def min_evs(name):
    out = []
    for ln in open(os.path.join(SMALL, name)):
        m = re.match(r"\s+- min ev: ([0-9.e+-]+)", ln)
        if m:
            out.append(float(m.group(1)))
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    data = json.load(open(os.path.join(here, "reference_small_outputs.json")))
    for fx in data["fixtures"]:
        got = min_evs(fx["file"])
        want = [v for v in fx["min_ev"] if v is not None]
        assert got == want, (fx["file"], got, want)
    table = {}
    for ln in open(os.path.join(REF, "subdivided_hyper_cube_balanced_01.output")):
        t = ln.split()
        table[t[0]] = [int(t[1]), int(t[2]), int(t[3]), int(t[4])]
    for k, v in data["mesh_sizes"].items():
        if not k.startswith("_"):
            assert table[k] == v, (k, table[k], v)
    print("fixtures match the reference files")
