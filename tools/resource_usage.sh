#!/bin/bash
# per-kernel register / spill / scratch figures of one .hip file: tools/resource_usage.sh emme_amd/csrc/linstep_blocked.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 |
python3 -c "
import sys,re,subprocess
cur=None; vals={}
for line in sys.stdin:
    m=re.search(r'remark: Function Name: (\S+)',line)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
        cur=re.sub(r'\(.*','',cur.replace('emme::(anonymous namespace)::','').replace('void ',''))
        vals={}
        continue
    m=re.search(r'remark:\s+([\w \[\]/]+?): (\w+)',line)
    if m and cur:
        vals[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith('LDS Size'):
            print(f\"{cur:58s} VGPR {vals.get('VGPRs','?'):>4} AGPR {vals.get('AGPRs','?'):>3} SGPR {vals.get('TotalSGPRs','?'):>4} spillV {vals.get('VGPRs Spill','?'):>4} spillS {vals.get('SGPRs Spill','?'):>4} scratch {vals.get('ScratchSize [bytes/lane]','?'):>5} occ {vals.get('Occupancy [waves/SIMD]','?')}\")
"
