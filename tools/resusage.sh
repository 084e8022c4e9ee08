#!/bin/bash
# Per-kernel register / LDS / occupancy table of slamhip.hip (compile only, no GPU).
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I include "${1:-slam_amd/csrc/slamhip.hip}" -o /tmp/_res.so -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
rows=[];cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass',l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=='Function Name':
        cur={'name':v}; rows.append(cur)
    else: cur[k.split()[0]]=v
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n).replace('void ','')
    print(f\"{n:48s} sgpr {r.get('TotalSGPRs'):>4s} vgpr {r.get('VGPRs'):>4s} scratch {r.get('ScratchSize'):>4s} occ {r.get('Occupancy'):>2s} lds {r.get('LDS')}\")
"
