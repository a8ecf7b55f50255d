"""Condense rocprofv3 outputs under gpurun_out/ into small tracked files under profiles/.
usage: python tools/prof_summary.py r01"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, out = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
os.makedirs(out, exist_ok=True)
st = glob.glob(os.path.join(go, 'prof_stats', '*', '*_kernel_stats.csv'))
if st:
    shutil.copy(st[0], os.path.join(out, f'{tag}_kernel_stats.csv'))
    print('wrote', f'profiles/{tag}_kernel_stats.csv')
rows = defaultdict(lambda: defaultdict(list))
for name, ctr in (('prof_fetch', 'FETCH_SIZE'), ('prof_write', 'WRITE_SIZE')):
    for f in glob.glob(os.path.join(go, name, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr:
                key = (r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'])
                rows[key][ctr].append(float(r['Counter_Value']))
                rows[key]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(os.path.join(out, f'{tag}_pmc_hbm_traffic.csv'), 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'grid_threads', 'wg', 'launches', 'avg_us(pmc run)', 'FETCH_SIZE_KB_avg', 'WRITE_SIZE_KB_avg',
                'hbm_MB_per_launch = (2*FETCH + WRITE)*1024/1e6  [gfx950: FETCH_SIZE counts 64 B per 128-B request]'])
    for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1]['dur_us'])):
        fe = sum(v['FETCH_SIZE']) / max(len(v['FETCH_SIZE']), 1)
        wr = sum(v['WRITE_SIZE']) / max(len(v['WRITE_SIZE']), 1)
        n = max(len(v['FETCH_SIZE']), len(v['WRITE_SIZE']))
        w.writerow([k[:110], g, wg, n, round(sum(v['dur_us']) / len(v['dur_us']), 1), round(fe, 1), round(wr, 1),
                    round((2 * fe + wr) * 1024 / 1e6, 2)])
print('wrote', f'profiles/{tag}_pmc_hbm_traffic.csv')
