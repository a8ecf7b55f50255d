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


def newest(pattern):
    """gpurun merges every call's outputs into the same gpurun_out/ directories: only the latest file of a pass counts."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


st = newest(os.path.join(go, 'prof_stats', '*', '*_kernel_stats.csv'))
if st:
    shutil.copy(st[0], os.path.join(out, f'{tag}_kernel_stats.csv'))
    print('wrote', f'profiles/{tag}_kernel_stats.csv')
# per (kernel, grid) durations from the kernel trace: one template instance serves several layers, the grid tells them apart
tr = newest(os.path.join(go, 'prof_stats', '*', '*_kernel_trace.csv'))
if tr:
    per = defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        key = (r['Kernel_Name'], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
        per[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    with open(os.path.join(out, f'{tag}_kernel_by_grid.csv'), 'w') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'grid_x', 'grid_y', 'grid_z', 'wg', 'launches', 'avg_us', 'min_us', 'max_us', 'total_ms'])
        for (k, gx, gy, gz, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if sum(v) < 50.:
                continue
            w.writerow([k[:120], gx, gy, gz, wg, len(v), round(sum(v) / len(v), 1), round(min(v), 1), round(max(v), 1),
                        round(sum(v) / 1e3, 3)])
    print('wrote', f'profiles/{tag}_kernel_by_grid.csv')
bj = os.path.join(go, 'prof_stats_bench.json')
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(out, f'{tag}_bench_under_rocprof.json'))
# PMC rows carry only the total grid size; several layers share it.  Every step issues the same dispatch sequence,
# so the j-th dispatch of a kernel (mod its dispatches per step) has the 3-D grid of the trace's j-th one.
seq3d, per_step = defaultdict(list), {}
if tr:
    n_steps = 0
    for r in csv.DictReader(open(tr[0])):
        seq3d[r['Kernel_Name']].append('x'.join((r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])))
        n_steps += 'adam_step' in r['Kernel_Name']
    per_step = {k: len(v) // max(n_steps, 1) for k, v in seq3d.items()}
rows = defaultdict(lambda: defaultdict(list))
for name, ctr in (('prof_fetch', 'FETCH_SIZE'), ('prof_write', 'WRITE_SIZE')):
    for f in newest(os.path.join(go, name, '*', '*_counter_collection.csv')):
        seen = defaultdict(int)
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id'])):
            if r['Counter_Name'] == ctr:
                k = r['Kernel_Name']
                j = seen[k]
                seen[k] += 1
                grid = seq3d[k][j % per_step[k]] if per_step.get(k) else r['Grid_Size']
                key = (k, grid, r['Workgroup_Size'])
                rows[key][ctr].append(float(r['Counter_Value']))
                rows[key]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(os.path.join(out, f'{tag}_pmc_hbm_traffic.csv'), 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'grid (threads, x*y*z from the trace run)', 'wg', 'launches', 'avg_us(pmc run)', 'FETCH_SIZE_KB_avg',
                'WRITE_SIZE_KB_avg',
                'hbm_MB_per_launch = (2*FETCH + WRITE)*1024/1e6  [gfx950: FETCH_SIZE counts 64 B per 128-B request]'])
    for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1]['dur_us'])):
        fe = sum(v['FETCH_SIZE']) / max(len(v['FETCH_SIZE']), 1)
        wr = sum(v['WRITE_SIZE']) / max(len(v['WRITE_SIZE']), 1)
        n = max(len(v['FETCH_SIZE']), len(v['WRITE_SIZE']))
        w.writerow([k[:110], g, wg, n, round(sum(v['dur_us']) / len(v['dur_us']), 1), round(fe, 1), round(wr, 1),
                    round((2 * fe + wr) * 1024 / 1e6, 2)])
print('wrote', f'profiles/{tag}_pmc_hbm_traffic.csv')

# SQ / TCC counter passes (collected WITH the persistent GRU scans on): per kernel template, averages per dispatch
for name, fname in (('prof_sq', 'pmc_sq'), ('prof_tcc', 'pmc_tcc'), ('prof_mfma', 'pmc_mfma')):
    files = newest(os.path.join(go, name, '*', '*_counter_collection.csv'))
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if 'rocclr' in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'][:110], r['Grid_Size'], r['Workgroup_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[key]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    counters = sorted({c for v in agg.values() for c in v if c != 'dur_us'})
    with open(os.path.join(out, f'{tag}_{fname}.csv'), 'w') as fh:
        w = csv.writer(fh)
        extra = (['wait_any_frac', 'wait_inst_frac', 'active_frac'] if fname == 'pmc_sq' else ['l2_hit_rate'] if fname == 'pmc_tcc'
                 else ['mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)', 'bf16_tflops_at_kernel_clock_equiv'])
        w.writerow(['kernel', 'grid_threads', 'wg', 'dispatches', 'avg_us(pmc run)'] + counters + extra)
        for (k, g, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]['dur_us'])):
            n = len(v[counters[0]])
            if sum(v['dur_us']) / max(len(counters), 1) < 200.:
                continue
            avg = {c: sum(v[c]) / max(len(v[c]), 1) for c in counters}
            if fname == 'pmc_sq':
                wc = max(avg.get('SQ_WAVE_CYCLES', 0.), 1.)
                ex = [round(avg.get('SQ_WAIT_ANY', 0.) / wc, 3), round(avg.get('SQ_WAIT_INST_ANY', 0.) / wc, 3),
                      round(avg.get('SQ_ACTIVE_INST_ANY', 0.) / wc, 3)]
            elif fname == 'pmc_tcc':
                ex = [round(avg.get('TCC_HIT_sum', 0.) / max(avg.get('TCC_HIT_sum', 0.) + avg.get('TCC_MISS_sum', 0.), 1.), 3)]
            else:
                clk = max(avg.get('GRBM_GUI_ACTIVE', 0.) / 8., 1.)                 # the kernel's duration in shader clocks
                dur = sum(v['dur_us']) / len(v['dur_us'])
                ex = [round(avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.) / (1024. * clk), 3),
                      round(avg.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.) * 512 / (dur * 1e-6) / 1e12, 1)]
            w.writerow([k, g, wg, n, round(sum(v['dur_us']) / len(v['dur_us']), 1)] + [round(avg[c], 1) for c in counters] + ex)
    print('wrote', f'profiles/{tag}_{fname}.csv')
for cfg in ('c3', 'c5', 'deep'):
    stc = newest(os.path.join(go, f'prof_stats_{cfg}', '*', '*_kernel_stats.csv'))
    if stc:
        shutil.copy(stc[0], os.path.join(out, f'{tag}_{cfg}_kernel_stats.csv'))
        print('wrote', f'profiles/{tag}_{cfg}_kernel_stats.csv')
    bjc = os.path.join(go, f'prof_stats_bench_{cfg}.json')
    if os.path.exists(bjc):
        shutil.copy(bjc, os.path.join(out, f'{tag}_{cfg}_bench_under_rocprof.json'))

# HBM bytes per launch of the layers bench.py may name as its roofline kernel (bench.py copies the value into
# `roofline.traffic`); keyed by bench.py's "<entry point> <layer tag>", matched to (kernel template, 3-D grid).
import json
ROOFLINE_ROWS = {
    'pbsed_conv_bwd_weight 128->128 k3x3 B32 F16 T500 x3pc': ('conv_wgrad_pc_kernel<2, 2, 1, 2, 2', 'x2x2'),      # grid (split, cin tiles, cout tiles)
    'pbsed_conv_bwd_weight 128->256 k3x3 B32 F8 T500 x3pc': ('conv_wgrad_pc_kernel<2, 2, 1, 2, 2', 'x2x4'),
    'pbsed_conv_bwd_data_winox3 128->128 k3x3 B32 F16 T500 winox3': ('conv_winox3_kernel<false, true, true, 64>', None),
    'pbsed_conv_fwd_winox3 128->128 k3x3 B32 F16 T500 winox3': ('conv_winox3_kernel<true, false, false, 64>', None),
    'pbsed_conv_bwd_weight 128->128 k3x3 B32 F16 T500 wino': ('conv_wgrad_wino_kernel', None),
    'pbsed_conv_bwd_data_wino 128->128 k3x3 B32 F16 T500 wino': ('conv_wino_kernel<false, true>', None),
    'pbsed_conv_fwd_wino 128->128 k3x3 B32 F16 T500 wino': ('conv_wino_kernel<true, false>', None),
}
traffic = {}
for tagname, (kern, grid) in ROOFLINE_ROWS.items():
    cands = [(kk, v) for kk, v in rows.items() if kern in kk[0] and (grid is None or kk[1] == grid or kk[1].endswith(grid)) and v['FETCH_SIZE'] and v['WRITE_SIZE']]
    cands.sort(key=lambda kv: -sum(kv[1]['dur_us']) / len(kv[1]['dur_us']))
    for (k, g, wg), v in cands[:1]:
        if True:
            fe = sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE'])
            wr = sum(v['WRITE_SIZE']) / len(v['WRITE_SIZE'])
            traffic[tagname] = {'hbm_bytes_per_launch': round((2 * fe + wr) * 1024), 'kernel': kern, 'grid': g,
                                'source': f'profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                          'bytes = (2*FETCH_SIZE + WRITE_SIZE) KB)'}
# The persistent scans do not run under the FETCH / WRITE passes (and HBM is not what they move): their `traffic` is the L2
# request count of the TCC pass (collected with the scans on) x 128 bytes - polls, saved-factor loads, input projections and
# output stores of one launch as the L2s saw them.
tcc_csv = os.path.join(out, f'{tag}_pmc_tcc.csv')
if os.path.exists(tcc_csv):
    for r in csv.DictReader(open(tcc_csv)):
        for key, name in (('gru_granule_fwd', 'gru_granule_fwd'), ('gru_granule_bwd', 'gru_granule_bwd')):
            if name in r['kernel'] and key not in traffic:
                traffic[key] = {'l2_request_bytes_per_launch': round(float(r['TCC_REQ_sum']) * 128), 'l2_hit_rate': float(r['l2_hit_rate']),
                                'kernel': r['kernel'][:60], 'grid': r['grid_threads'],
                                'source': f'profiles/{tag}_pmc_tcc.csv (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum, own pass, '
                                          'bytes = TCC_REQ_sum x 128: a request moves at most one 128-byte line)'}
if traffic:
    json.dump(traffic, open(os.path.join(out, 'roofline_traffic.json'), 'w'), indent=1)
    print('wrote profiles/roofline_traffic.json', list(traffic))

# per-kernel counter sets of tools/prof_kernel.sh (FETCH / WRITE / TCC / SQ / MFMA passes of ONE kernel's own bench command)
for pk, name in (('pk_wx', 'winox3'), ('pk_wgpc', 'wgrad_pc'), ('pk_c1', 'conv1d_pc'), ('pk_gw', 'gru_wgrad_pc'), ('pk_lm', 'logmel')):
    f = os.path.join(go, pk, 'summary.json')
    if os.path.exists(f):
        shutil.copy(f, os.path.join(out, f'{tag}_pmc_{name}.json'))
        print('wrote', f'profiles/{tag}_pmc_{name}.json')
