"""Per-kernel time per training step from a rocprofv3 --kernel-trace CSV, over the steps between the first and the last
AdamW launch only (the warm-up step carries one-off costs - first-touch page faults, code loads - that the --stats summary
averages in): usage  trace_step_table.py <kernel_trace.csv> > profiles/rNN_bench_kernel_steps.txt"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
assert len(ad) >= 3, 'need at least three optimizer steps in the trace'
lo, hi, nsteps = ad[0] + 1, ad[-1] + 1, len(ad) - 1
sel = rows[lo:hi]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\((anonymous namespace::)?\w+Params\)$', '', n)
    n = re.sub(r'^at::native::', '', n)
    return n.split('(')[0][:72]


tot = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = short(r['Kernel_Name'])
    tot[k][0] += 1
    tot[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6 / nsteps
ksum = sum(v[1] for v in tot.values()) / 1e6 / nsteps
print(f'# {nsteps} steps between the first and the last AdamW launch of the trace: wall {wall:.2f} ms/step, kernel time {ksum:.2f} ms/step')
print(f'{"kernel":74s} {"n/step":>7s} {"ms/step":>8s} {"avg us":>8s}')
for k, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:74s} {n / nsteps:7.1f} {ns / 1e6 / nsteps:8.3f} {ns / n / 1e3:8.1f}')
