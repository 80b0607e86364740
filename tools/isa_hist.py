#!/usr/bin/env python3
"""dev helper: instruction histogram of the innermost loop(s) of one kernel of a hipcc -S listing.
usage: isa_hist.py listing.s 'substring of the demangled kernel name'"""
import re, subprocess, sys, collections
s = open(sys.argv[1]).read()
names = re.findall(r'^(_ZN3smg\S+):', s, re.M)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
target = [n for n, d in zip(names, dem) if sys.argv[2] in d][0]
i = s.index('\n' + target + ':')
b = s[i:s.index('.Lfunc_end', i)].split('\n')
labels = [(n, l) for n, l in enumerate(b) if re.match(r'\.LBB\d+_\d+:', l.strip())]
inl = [n for n, l in labels if 'Loop' in l]
if inl:
    after = [n for n, l in labels if n > inl[-1]]
    end = after[0] if after else len(b)
    loop = [l for l in b[inl[0]:end] if l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(l.split()[0] for l in loop)
    valu = sum(n for a, n in c.items() if a.startswith('v_') and not a.startswith('v_mfma'))
    print(f'{len(loop)} instructions in loop blocks, {sum(n for a,n in c.items() if a.startswith("v_mfma"))} mfma, {valu} valu')
    print('   ' + ', '.join(f'{n} {a}' for a, n in c.most_common(50)))
