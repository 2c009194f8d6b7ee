#!/usr/bin/env python3
"""print the last N kernel records of a rocprofv3 kernel trace (csv) as a timeline: tools/trace_tail.py trace.csv [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
span = rows[-n:]
t0 = int(span[0]["Start_Timestamp"])
for r in span:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("asmamd::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:46]
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} q{r['Queue_Id']} grid {r['Grid_Size_X']:>8} {name}")
