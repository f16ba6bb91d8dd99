#!/usr/bin/env python3
"""Print the per-launch timeline of the last bench step from a rocprofv3 kernel trace csv (dev tool)."""
import csv, sys, glob
path = sys.argv[1]
f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'aefft' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'r2c_rows' in r['Kernel_Name']]
s = idx[-1] if idx else 0
t0 = int(rows[s]['Start_Timestamp']); tot = 0
agg = {}
for r in rows[s:]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d
    name = r['Kernel_Name'].replace('aefft::', '').replace('void ', '')
    name = name[:name.index('(')] if '(' in name else name
    agg[name] = agg.get(name, 0) + d
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us {d/1e3:8.1f}us grid={r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']}x{r['Workgroup_Size_Y']} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']} {name[:60]}")
print('sum kernel us', tot / 1e3, 'span us', (int(rows[-1]['End_Timestamp']) - t0) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:9.1f}us  {k}")
