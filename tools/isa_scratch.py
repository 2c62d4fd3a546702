"""Where a kernel's scratch (spill) traffic sits relative to its loops: for every kernel in a hipcc -S listing, the scratch
loads / stores with the loop depth of their basic block.  A scratch_load inside an MFMA K loop comes with s_waitcnt vmcnt(0)
(one exposed memory latency per K-step): the thing to look for after any change to the GEMM kernels.
usage: isa_scratch.py file.s [substring-of-kernel-name]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cur, depth, has_mfma = None, 0, {}
rows = {}
for l in src:
    m = re.match(r'^(_Z\S+):', l)
    if m:
        cur, depth = m.group(1), 0
        rows[cur] = []
        continue
    if cur is None:
        continue
    if l.startswith('.Lfunc_end'):
        cur = None
        continue
    m = re.match(r'^\.LBB\S+:\s*;?(.*)', l)
    if m:
        d = re.search(r'Depth=(\d+)', m.group(1))
        depth = int(d.group(1)) if d else 0
        blk = l.split(':')[0]
        continue
    t = l.strip()
    if t.startswith('scratch_'):
        rows[cur].append((depth, t.split(';')[0].strip()))
for k, v in rows.items():
    if pat not in k or 'kernel' not in k:
        continue
    by = {}
    for d, t in v:
        by.setdefault(d, [0, 0])['load' in t] += 1
    print(k[:110], ' | '.join(f'depth {d}: {n[1]} loads {n[0]} stores' for d, n in sorted(by.items())) or 'no scratch')
