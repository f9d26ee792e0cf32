#!/usr/bin/env python3
"""Basic-block summary of one kernel in a hipcc --save-temps .s file: label, instruction count, counts of the ops that matter.
usage: tools/asm_blocks.py file.s mangled_kernel_name_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = [l.split(':')[0] for l in s.split('\n') if ':' in l and sys.argv[2] in l.split(':')[0] and l[0] not in '.; \t'][0]
i = s.index(name + ':')
j = s.index('s_endpgm', i)
blocks = []
cur = ['entry', []]
for l in s[i:j].split('\n')[1:]:
    t = l.strip()
    m = re.match(r'^(\.?[A-Za-z0-9_$.]+):', t)
    if not t or (t.startswith((';', '//', '.')) and not m):
        continue
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
    else:
        cur[1].append(t)
blocks.append(cur)
PRE = ('ds_', 'v_log', 'v_exp', 'global', 'v_mul_f64', 'v_fma', 'v_add_f64', 's_cbranch', 'v_cmp', 'v_cndmask', 'v_cvt', 'scratch', 'v_pk')
tot = 0
for b in blocks:
    ops = collections.Counter(x.split()[0] for x in b[1])
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    key = {k: v for k, v in ops.items() if k.startswith(PRE)}
    print("%-10s n=%4d valu=%4d %s" % (b[0], len(b[1]), valu, key))
