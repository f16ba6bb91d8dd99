#!/usr/bin/env python3
import csv, glob, json, sys
from collections import defaultdict
path = sys.argv[1]
order = json.load(open('gpurun_out/sweep_order.json'))
f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'aefft' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps = []; cur = None
for r in rows:
    if 'r2c_rows_kernel<512>' in r['Kernel_Name']:
        cur = []; steps.append(cur)
    if cur is not None and 'contract' in r['Kernel_Name']:
        n = r['Kernel_Name']; n = n[n.index('contract'):n.index('(')] if '(' in n else n
        cur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp']), n))
assert len(steps) == len(order), (len(steps), len(order))
best = defaultdict(dict)
for st, cfg in zip(steps, order):
    for j, (d, n) in enumerate(st):
        k = (cfg, n)
        best[j][k] = min(best[j].get(k, 1 << 60), d)
tot = 0
for j in sorted(best):
    items = sorted(best[j].items(), key=lambda kv: kv[1])
    auto = [v for (c, n), v in best[j].items() if c == 'auto'][0]
    tot += items[0][1]
    print(f"#{j:2d} auto {auto/1e3:6.1f}us | " + "  ".join(f"{c}:{v/1e3:.1f}" for (c, n), v in items[:6]))
print("sum of best", tot / 1e3, "us; sum auto", sum([v for j in best for (c, n), v in best[j].items() if c == 'auto']) / 1e3)
