#!/bin/bash
# usage: tools/kernel_resources.sh prnn   -> compact VGPR/AGPR/scratch/occupancy table for one TU
f=${1:-prnn}
VF="-mllvm -amdgpu-mfma-vgpr-form"; [ "$f" = split_stream ] && VF=""
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -ffp-contract=fast $([ "$f" = split -o "$f" = split_stream ] && echo -fno-slp-vectorize) $VF \
  -Rpass-analysis=kernel-resource-usage -c rnnwavefunctions_amd/csrc/$f.hip -o /tmp/$f.o 2>&1 |
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: 
        m2=re.search(r": +(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|VGPR Spill|SGPR Spill): (.*?) \[", line)
        if not m2: continue
        k,v=m2.group(1),m2.group(2)
    else:
        kv=m.group(1).split(": ")
        if len(kv)<2: continue
        k,v=kv[0].strip(),kv[1].strip()
    if k in ("Function Name","Name"):
        cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip().split("(")[0]
    print("%-62s V=%-4s A=%-4s S=%-4s scratch=%-4s occ=%s" % (n[-62:], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
'
