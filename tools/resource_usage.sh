#!/bin/bash
# dev helper: VGPR / AGPR / SGPR / scratch / occupancy / LDS of every kernel of the three translation units
# (hipcc -Rpass-analysis=kernel-resource-usage); `tools/resource_usage.sh > profiles/resource_usage_rNN.txt`
cd "$(dirname "$0")/../smg-multimodal-grasping_amd/csrc"
for f in forward backward engine; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c $f.hip -o /dev/null 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur)
    for key,pat in [('V',r' VGPRs: (\d+)'),('A',r'AGPRs: (\d+)'),('scr',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('S',r' SGPRs: (\d+)')]:
        m=re.search(pat,l)
        if m and cur is not None: cur[key]=m.group(1)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
print('== $f.hip')
for r,n in zip(rows,names):
    n=n.replace('smg::','').replace('GemmCfg','Cfg')
    n=re.sub(r'\(.*','',n)
    print(n[:130].ljust(130),'V',r.get('V'),'A',r.get('A'),'S',r.get('S'),'scr',r.get('scr'),'occ',r.get('occ'),'lds',r.get('lds'))
"
done
