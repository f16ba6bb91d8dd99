cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/xe_cfg2 -o g -- python3 $GRAFT_REPO_ROOT/tools/cfgstep.py cfg2 200 > $GRAFT_REPO_ROOT/gpurun_out/xe_cfg2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/xe_cfg2/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'aefft' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'r2c_rows' in r['Kernel_Name']]
k = len(idx) // 2; s = idx[k]; e = idx[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = r['Kernel_Name'].replace('aefft::', '').replace('void ', ''); n = n[:n.index('(')] if '(' in n else n
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f}us {d/1e3:7.1f}us grid={r['Grid_Size_X']},{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']} {n[:50]}")
print('period', (int(rows[e]['Start_Timestamp']) - t0) / 1e3)
PY
grep cfg2 gpurun_out/xe_cfg2.log
