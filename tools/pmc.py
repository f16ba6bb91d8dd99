#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_traffic.json: per kernel family, launches and average HBM-side bytes per launch.
gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B -> doubled; WRITE_SIZE is exact; both are in KiB."""
import csv, glob, json, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d, name):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name or 'aefft' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].replace('void ', '').replace('aefft::', '')
        k = k[:k.index('<')] if '<' in k else k[:k.index('(')]
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
    return acc
F, W = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
res = {}
for k in sorted(set(F) | set(W)):
    n = max(F[k][0], W[k][0])
    res[k] = {"launches": n, "fetch_bytes_per_launch": 2 * F[k][1] * 1024 / max(F[k][0], 1), "write_bytes_per_launch": W[k][1] * 1024 / max(W[k][0], 1)}
    res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
json.dump(res, open(out, 'w'), indent=1)
for k, v in res.items():
    print(f"{k:28s} n={v['launches']:5d} fetch={v['fetch_bytes_per_launch']/1e6:8.2f} MB write={v['write_bytes_per_launch']/1e6:8.2f} MB")
