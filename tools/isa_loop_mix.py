"""Instruction mix of the hottest loop (the one holding the most MFMAs, or
VALU when there are none) of one kernel in a gfx950 assembly listing:
    hipcc ... --cuda-device-only -S x.hip -o x.s
    python tools/isa_loop_mix.py x.s <substring of the mangled kernel name>"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = None
for mm in re.finditer(r'^(\S+):\s*; @\S+\n(.*?)s_endpgm', text, re.S | re.M):
    if key in mm.group(1):
        m = mm
        break
if m is None:
    sys.exit("kernel not found")
print(m.group(1))
lines = [l.strip() for l in m.group(2).split('\n')]
lines = [l.split(';')[0].strip() for l in lines]
lines = [l for l in lines if l and (l.endswith(':') or not l.startswith(('.', '//')))]
best = None
for i, l in enumerate(lines):
    if not l.endswith(':'):
        continue
    name = re.escape(l[:-1])
    for k in range(i + 1, len(lines)):
        if re.match(r's_c?branch\w*\s+' + name + r'\b', lines[k]):
            seg = lines[i:k + 1]
            score = (sum('mfma' in x for x in seg), len(seg))
            if best is None or score > best[0]:
                best = (score, i, k)
_, i, k = best
seg = [x for x in lines[i:k + 1] if not x.endswith(':')]
c = collections.Counter()
for x in seg:
    op = x.split()[0]
    if 'mfma' in op: c['mfma'] += 1
    elif op.startswith('ds_'): c[op] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_waitcnt'): c['s_waitcnt'] += 1
    elif op.startswith('s_nop'): c['s_nop'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    else: c[op] += 1
print(len(seg), "instructions in the loop:", dict(c))
vc = collections.Counter(x.split()[0] for x in seg if x.startswith('v_'))
print(vc.most_common(30))
