"""Summarise the passes of tools/pmc.sh: per kernel (short name), mean counter values per dispatch and the derived
ratios used in DESIGN.md.  HBM bytes follow MI355X_MICROARCH.md (HBM section): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
usage: pmc_summary.py <dir with pass*/> <out.json> [failed pass numbers ...]   (listed under "_failed_passes")"""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r'(gemm_nt2_kernel|gemm_tn2_kernel)<([^>]*)>', name)
    if m:
        args = [a.strip() for a in m.group(2).split(',')]
        return '%s<%s>' % (m.group(1).replace('_kernel', ''), ','.join(args[:4] if 'nt2' in m.group(1) else args[:2]))
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.search(r'(\w+_kernel\w*(?:<[^>(]*>)?|\w+)', name.split('(')[0].split('::')[-1])
    return m.group(1).replace(' ', '') if m else name[:40]


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f'{root}/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[short(r['Kernel_Name'])][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f'{root}/pass1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[short(r['Kernel_Name'])]
        d[0] += 1
        d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
res = {}
for k, cs in acc.items():
    c = {n: v[1] / v[0] for n, v in cs.items()}
    d = {'dispatches': max(v[0] for v in cs.values()), 'counters_mean_per_dispatch': c}
    if k in dur:
        d['avg_us_under_pmc_pass1'] = dur[k][1] / dur[k][0]
    wc = c.get('SQ_WAVE_CYCLES')
    if wc:
        d['per_wave_cycle'] = {n.lower().replace('sq_', ''): round(c[n] / wc, 3) for n in
                               ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU',
                                'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS') if n in c}
    if c.get('SQ_INSTS_MFMA'):
        d['valu_per_mfma'] = round(c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA'], 2)
        d['lds_per_mfma'] = round(c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA'], 2)
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        # SQ_BUSY_CYCLES is summed over the 8 XCDs' SQs (per-SE granularity varies): report the raw ratio only
        d['mfma_busy_over_sq_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES'], 3)
    if c.get('TCC_HIT_sum') is not None and c.get('TCC_MISS_sum') is not None and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
        d['l2_hit_rate'] = round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 3)
    if 'FETCH_SIZE' in c or 'WRITE_SIZE' in c:
        d['hbm_bytes_per_launch_corrected'] = (2 * c.get('FETCH_SIZE', 0.0) + c.get('WRITE_SIZE', 0.0)) * 1024
    if c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict_frac'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'], 3)
    res[k] = d
if len(sys.argv) > 3:
    res['_failed_passes'] = [int(x) for x in sys.argv[3:]]
json.dump(res, open(out, 'w'), indent=1)
for k, d in res.items():
    if isinstance(d, dict):
        print(k, {x: d[x] for x in d if x not in ('counters_mean_per_dispatch',)})
