#!/bin/bash
# usage: tools/resusage.sh vtc_amd/csrc/file.hip [extra hipcc flags]  -> per kernel: VGPRs, SGPRs, spills, scratch
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c "$f" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
name=None; row={}
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        if name: print(name,row)
        name=m.group(1); row={}
    for k in ('VGPRs','SGPRs Spill','VGPRs Spill','ScratchSize','TotalSGPRs'):
        m=re.search(r'    '+k+r'(?: \[bytes/lane\])?: (\d+)',l)
        if m: row[k]=int(m.group(1))
if name: print(name,row)
"
