"""Register / spill table of one object's kernels from its -Rpass-analysis=kernel-resource-usage report.
  python tools/ru_table.py diffusion_amd/csrc/gemm_nt_v2.ru.txt [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
PATS = {'vgpr': r'VGPRs: (\d+)', 'spill': r'VGPRs Spill: (\d+)', 'sgpr': r'SGPRs: (\d+)',
        'scratch': r'ScratchSize \[bytes/lane\]: (\d+)', 'occ': r'Occupancy \[waves/SIMD\]: (\d+)'}
for b in re.split(r'(?=remark: [^\n]*Function Name)', txt):
    m = re.search(r'Function Name: (\S+)', b)
    if not m:
        continue
    d = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    d = d.replace('(anonymous namespace)::', '').replace('void ', '')[:100]
    if flt in d:
        vals = ' '.join(f'{k}={(re.search(pat, b) or [None, "?"])[1]:>3s}' for k, pat in PATS.items())
        print(f'{d:100s} {vals}')
