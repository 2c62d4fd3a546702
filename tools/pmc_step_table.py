"""Per-kernel table from the passes of `bash tools/pmc.sh step python3 bench.py ... --steps 1 --warmup 1`: launches, average duration,
VALU-active / active / wait fractions of the wave cycles, VALU instructions per vector-memory instruction, and the fabric rate
(2*FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md, Infinity-Cache hits included) of every kernel of a training step.
usage: pmc_step_table.py gpurun_out/pmc/step > profiles/rNN_pmc_step_kernels.txt"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]


def short(n):
    m = re.search(r'(\w+_kernel\w*)(<[^(]*>)?', n)
    if not m:
        return n[:40]
    name, t = m.group(1), (m.group(2) or '').replace(' ', '')
    if name == 'gemm_nt2_kernel':
        a = t.strip('<>').split(',')
        return 'nt2<%s|ups%s|geglu%s|early%s|persist%s|de%s%s>' % (','.join(a[:4]), a[5][0], a[6], a[7][0], a[9][0] if len(a) > 9 else '?',
                                                             a[10][0] if len(a) > 10 else 'f', a[11][0] if len(a) > 11 else 'f')
    return name + t[:24]


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
rows = []
for k, cs in acc.items():
    c = {n: v[1] / v[0] for n, v in cs.items()}
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if not wc or k not in dur:
        continue
    avg = dur[k][1] / dur[k][0]
    rows.append((dur[k][1], k, dur[k][0], avg, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                 c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc,
                 c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_VMEM', 1), 1),
                 (2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / (avg * 1e-6) / 1e12))
rows.sort(reverse=True)
print('# one bench.py step pair (--steps 1 --warmup 1) under rocprofv3 --pmc, separate passes (tools/pmc.sh); sorted by total time')
print(f'{"kernel":52s} {"n":>5s} {"avg us":>8s} {"valu":>6s} {"active":>6s} {"wait":>6s} {"winst":>6s} {"V/VMEM":>7s} {"TB/s":>6s}')
for r in rows[:40]:
    print(f'{r[1][:52]:52s} {r[2]:5d} {r[3]:8.1f} {r[4]:6.3f} {r[5]:6.3f} {r[6]:6.3f} {r[7]:6.3f} {r[8]:7.1f} {r[9]:6.2f}')
