#!/usr/bin/env python3
"""dev helper: condensed ISA (waitcnt / memory ops / branches, MFMA and VALU counts) of one kernel of a hipcc -S listing.
usage: isa_loop.py listing.s 'substring of the demangled kernel name'"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
names = re.findall(r'^(_ZN3smg\S+):', s, re.M)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
target = [n for n, d in zip(names, dem) if sys.argv[2] in d][0]
i = s.index('\n' + target + ':')
body = s[i:s.index('.Lfunc_end', i)].split('\n')
mf = va = 0
def flush():
    global mf, va
    if mf or va: print(f'      [{mf} mfma, {va} valu]')
    mf = va = 0
for l in body:
    t = l.strip()
    if t.startswith('v_mfma'): mf += 1; continue
    if t.startswith('v_'): va += 1; continue
    if re.match(r'(s_waitcnt|global_load|buffer_load|s_barrier|s_cbranch|s_branch|\.LBB|ds_write|ds_read|global_store|global_atomic|scratch_)', t):
        flush(); print(t.split('//')[0][:100])
flush()
