"""Instruction histogram of one kernel in a hipcc -S listing: python tools/isa_count.py file.s kernel_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ('' if len(sys.argv) < 4 else ''))
i = s.index('\n', s.index(sys.argv[2]))
f = s[i:s.index('.end_amdhsa_kernel', i)]
lines = f.split('\n')
ins = []
for x in lines:
    t = x.strip()
    if not x.startswith('\t') or not t or t.startswith(('.', ';')):
        continue
    ins.append(t.split()[0])
c = Counter(ins)
print(len(ins), 'instructions; valu', sum(n for k, n in c.items() if k.startswith('v_')))
print(sorted(c.items(), key=lambda x: -x[1])[:24])
# loops
labels = {}
for k, l in enumerate(lines):
    m = re.match(r'(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = k
for k, l in enumerate(lines):
    m = re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        body = [x.strip().split()[0] for x in lines[labels[m.group(1)]:k] if x.startswith('\t') and x.strip() and not x.strip().startswith(('.', ';'))]
        cb = Counter(body)
        print('loop', m.group(1), len(body), 'instr; valu', sum(n for q, n in cb.items() if q.startswith('v_')),
              'scratch', sum(n for q, n in cb.items() if 'scratch' in q), sorted(cb.items(), key=lambda x: -x[1])[:10])
