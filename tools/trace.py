#!/usr/bin/env python3
"""Print the per-launch timeline of one bench step from a rocprofv3 kernel trace csv (dev tool).
   tools/trace.py <dir> [1 = list the launches] -- the step in the MIDDLE of the trace (inside the timed loop: the last steps of a
   bench run belong to the profiled roofline pass, which runs serially with event brackets); period = start of the next step."""
import csv, sys, glob
path = sys.argv[1]
f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'aefft' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'r2c_rows' in r['Kernel_Name']]
k = len(idx) // 2 if len(idx) > 2 else max(len(idx) - 1, 0)
s = idx[k] if idx else 0
e = idx[k + 1] if idx and k + 1 < len(idx) else len(rows)
t0 = int(rows[s]['Start_Timestamp']); tot = 0
agg = {}
for r in rows[s:e]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
    name = r['Kernel_Name'].replace('aefft::', '').replace('void ', '')
    name = name[:name.index('(')] if '(' in name else name
    agg[name] = agg.get(name, 0) + d
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us {d/1e3:8.1f}us grid={r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']}x{r['Workgroup_Size_Y']} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']} {name[:60]}")
end = max(int(r['End_Timestamp']) for r in rows[s:e])
period = (int(rows[e]['Start_Timestamp']) - t0) / 1e3 if e < len(rows) else float('nan')
print(f"step {k} of {len(idx)}: sum of kernel durations {tot/1e3:.1f} us (side-stream kernels overlap the others), first start -> last end {(end-t0)/1e3:.1f} us, period to the next step {period:.1f} us")
for kk, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:9.1f}us  {kk}")
