#!/bin/bash
# development: register / scratch / occupancy of every kernel of the library as the compiler reports them (no GPU needed)
cd "$(dirname "$0")/../multiagent_planning_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -S --cuda-device-only -Rpass-analysis=kernel-resource-usage $KRES_DEFS -o /dev/null dmpc_api.hip 2>&1 |
python3 -c "
import sys,re
cur=None;rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)};rows.append(cur);continue
    m=re.search(r'remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)',ln)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    if '${1:-}' and '${1:-}' not in n: continue
    print(f\"{n[:70]:70s} VGPR {r.get('VGPRs',0):3d} AGPR {r.get('AGPRs',0):3d} SGPR {r.get('TotalSGPRs',0):3d} scratch {r.get('ScratchSize [bytes/lane]',0):4d} occ {r.get('Occupancy [waves/SIMD]',0)} sgpr-spill {r.get('SGPRs Spill',0)} vgpr-spill {r.get('VGPRs Spill',0)}\")
"
