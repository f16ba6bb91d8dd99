#!/usr/bin/env python3
"""dev tool: print name / calls / average us of the aefft kernels from rocprofv3 kernel_stats csv files (side by side)."""
import csv, re, sys
tabs = []
for p in sys.argv[1:]:
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"]
        if "aefft" not in n: continue
        n = re.sub(r"\(.*", "", n).replace("void ", "").replace("aefft::", "")
        d[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    tabs.append(d)
names = sorted(set().union(*tabs), key=lambda n: -max(t.get(n, (0, 0))[1] for t in tabs))
for n in names:
    print(f"{n:45s}" + "".join(f"{t.get(n, (0, 0))[1]:10.2f}" for t in tabs))
